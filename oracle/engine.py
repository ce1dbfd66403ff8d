"""Oracle compute engine: one nonlinear solve = the reference's ``self.solver.solve()``
(/root/reference/thermalporous/thermalmodel.py:165): SNES newtonls with the `basic` line search
Firedrake selects by default, right-preconditioned FGMRES, composite CPR/CPTR preconditioner.
Test infrastructure only (see oracle/__init__.py).

Same interface as thermalporous_amd.engine.HipEngine so that the host time loop
(thermalporous_amd/thermalmodel.py) can be exercised on CPU in tests by injection.
"""
import numpy as np

from .tpfa import Problem
from . import linalg as la

# SNES / KSP reason codes (PETSc numbering)
SNES_CONVERGED_FNORM_ABS = 2
SNES_CONVERGED_FNORM_RELATIVE = 3
SNES_CONVERGED_SNORM_RELATIVE = 4
SNES_DIVERGED_LINEAR_SOLVE = -3
SNES_DIVERGED_FNORM_NAN = -4
SNES_DIVERGED_MAX_IT = -5

DEFAULT_OPTS = dict(
    pc="cpr", decoup="No",
    ksp_rtol=1e-7, ksp_atol=1e-50, ksp_max_it=200, ksp_restart=200,
    snes_rtol=1e-8, snes_atol=1e-50, snes_stol=1e-8, snes_max_it=15,
    amg_omega=0.9,          # damped-Jacobi weight (round 3: 0.8 -> 0.9 buys 3 % fewer Krylov iterations on C4 at equal cycle cost, +4 % Newton steps/s
                            # over 80 time steps, measured twice; 0.88-0.9 is a plateau, 0.95 starts to fail solves, 1.0 loses 40 %; C1-C3 neutral)
    amg_min_cells=64, amg_nu=2, amg_full_levels=3, amg_coarse_pre=0, amg_coarse_post=1, amg_mid_skip=True, amg_tail_post=2, amg_single=False,
    schur_a11=False,
    fs_additive=False,      # pc_fieldsplit_type additive on (p,T): pc_fieldsplit_diag (singlephase.py:371-375)
    schur_selfp=False,      # pc_fieldsplit_schur_precondition selfp (pc_fieldsplit_selfp, singlephase.py:322-330)
    amg_dom_tau=0.25,       # relaxation-only truncation of diagonally dominant AMG hierarchies (oracle/linalg.py:SemiAMG)
    amg_gather_cells=600000,      # GPU multi-slab execution detail (same algebra): ignored here
    ilu_tile=None,          # None: (whole line, 8, 8) in 3-D, (whole line, 32, 1) in 2-D -- the GPU engine's default
    ilu_levels=0,           # sub_1_sub_pc_factor_levels: 0 or 1 (oracle/linalg.py:TiledILU1)
    bjacobi_blocks=None,    # -sub_1_pc_bjacobi_blocks: N boxes over the grid (same rule as the GPU engine, no lane limit)
    ilu_whole=False,        # one bjacobi block per slab (the GPU engine's whole-slab ILU(0))
)


def default_ilu_tile(n, nslabs=1, ncu=256):
    """(same rule as thermalporous_amd.engine.default_ilu_tile; tests/test_host_logic.py checks that they agree)
    bjacobi tile (t0, t1, t2) for a grid of internal extents n = (n0, n1, n2) cut into `nslabs` slabs along axis 2.
    Whole axis-0 lines always.  2-D: 32 columns (measured on C3 60x220: 64-wide tiles cost 123 wavefront steps for 60
    cells of depth, 32-wide ones 91 steps and +0.5 % Krylov iterations).  3-D: the t1 x t2 (32..64 columns, each side
    4..16) that minimises the sweep time of the busiest CU: one wavefront = one CU streams a tile's
    (n0 + t1 + t2 - 2) steps x t1*t2 lanes of factor data at the per-CU HBM rate, and `ncu` CUs work at a time --
    cost = ceil(tiles / ncu) * steps * lanes * (1 + |t1 - t2| / 100)  (elongated tiles cut more couplings per cell);
    ties go to the larger tile.  C4 (85 x 60 x 220): 6 x 9 -> 250 full tiles on 256 CUs, 54 lanes x 98 steps, instead of
    224 tiles of 8 x 8 (64 lanes x 99 steps, the 8th tile across half empty): 17 % fewer bytes through the busiest CU."""
    n0, n1, n2 = (int(v) for v in n)
    if n2 == 1:
        # 2-D sweeps are bound by their NUMBER OF STEPS (t0 + t1 - 1 dependent wavefront steps of ~0.3 us, a handful of
        # waves on the whole chip), not by bytes: cutting the lines into pieces of ~24 cells makes C1 (400 x 400) 67 %
        # faster at +8 % Krylov iterations (71 -> 119 Newton steps/s) and C3 (60 x 220) 12 % faster at equal counts
        return (-(-n0//max(1, -(-n0//24))), 32, 1)
    n2l = -(-n2//max(1, int(nslabs)))
    best = None
    for t1 in range(min(4, n1), min(16, n1) + 1):
        for t2 in range(min(4, n2l), min(16, n2l) + 1):
            lanes = t1*t2
            if lanes > 64 or (lanes < 32 and (t1 < min(16, n1) or t2 < min(16, n2l))):
                continue
            tiles = -(-n1//t1)*-(-n2l//t2)
            cost = -(-tiles//ncu)*(n0 + t1 + t2 - 2)*lanes*(1.0 + 0.01*abs(t1 - t2))
            key = (cost, -lanes, abs(t1 - t2))
            if best is None or key < best[0]:
                best = (key, (1 << 30, t1, t2))
    if best is None:
        return (1 << 30, min(n1, 8), min(n2l, 8))
    return best[1]


def blocks_to_tile(n, nblocks):
    """``-sub_1_pc_bjacobi_blocks N`` (tests/test_homo_wells.py:112,125 of the reference) -> the tile that cuts the grid
    into N boxes, whole axis-0 lines first, then the most compact box (the GPU engine applies the same rule plus its
    64-column limit per tile)."""
    n0, n1, n2 = (int(v) for v in n)
    best = None
    for k2 in range(1, min(n2, nblocks) + 1):
        if nblocks % k2:
            continue
        rem = nblocks//k2
        for k1 in range(1, min(n1, rem) + 1):
            if rem % k1 or rem//k1 > n0:
                continue
            k0 = rem//k1
            t = (-(-n0//k0), -(-n1//k1), -(-n2//k2))
            if (-(-n0//t[0]), -(-n1//t[1]), -(-n2//t[2])) != (k0, k1, k2):
                continue
            score = (k0 != 1, t[0]*t[1] + t[1]*t[2] + t[0]*t[2])
            if best is None or score < best[0]:
                best = (score, t)
    if best is None:
        raise ValueError("no tiling of %r into %d boxes" % (n, nblocks))
    return best[1]


class OracleEngine:
    def __init__(self, spec, opts=None):
        self.spec = spec
        self.opts = dict(DEFAULT_OPTS)
        self.opts.update(opts or {})
        self.prob = Problem(spec)
        self.b = self.prob.b
        self.u = None
        if self.opts.get("bjacobi_blocks") is not None:
            self.opts["ilu_tile"] = blocks_to_tile(spec["n"], self.opts["bjacobi_blocks"])
        if self.opts.get("ilu_whole"):           # one bjacobi block per slab: ILU(0) of the whole slab
            self.opts["ilu_tile"] = (1 << 30, 1 << 30, 1 << 30)
        if self.opts.get("ilu_tile") is None:
            self.opts["ilu_tile"] = default_ilu_tile(spec["n"], nslabs=int(self.opts.get("nslabs", 1)))
        self.pc = la.TwoStagePC(self.prob, self.opts)
        self.last = {}

    # state ------------------------------------------------------------------------------
    def set_state(self, u):
        self.u = self.prob.as_fields(np.array(u, dtype=float)).copy()

    def get_state(self):
        return self.u.copy()

    def set_old(self, u=None):
        self.prob.set_old(self.u if u is None else np.array(u, dtype=float))

    def set_dt(self, dt):
        self.prob.set_dt(dt)

    def get_old_state(self):
        return self.prob.u_old.copy()

    def restore_state(self):
        self.u = self.prob.u_old.copy()

    def saturation_range(self):
        return float(self.u[2].min()), float(self.u[2].max())

    def clamp_saturation(self):
        self.u[2] = np.clip(self.u[2], 0.0, 1.0)

    # pieces, exposed for parity tests ------------------------------------------------------
    def residual(self, u=None):
        return self.prob.residual(self.u if u is None else u)

    def jacobian(self, u=None, want_schur=False):
        return self.prob.jacobian(self.u if u is None else u, want_schur=want_schur)

    def well_rates(self, u=None):
        u = self.prob.as_fields(self.u if u is None else u)
        if self.prob.src is None:
            return {}
        c = self.prob.src["cell"]
        p, T, S = self.prob.split(u)
        args = [x.reshape(-1)[c] if x is not None else None for x in (p, T, S)]
        _, rates = self.prob.source_terms(*args, return_rates=True)
        return rates

    # the hot path --------------------------------------------------------------------------
    def linear_solve(self, J, Sm, F):
        o = self.opts
        self.pc.setup(J, Sm)
        return la.fgmres(lambda x: la.spmv_block(J, x), self.pc.apply, F, rtol=o["ksp_rtol"], atol=o["ksp_atol"],
                         restart=o["ksp_restart"], maxit=o["ksp_max_it"])

    def newton_solve(self):
        o = self.opts
        u = self.u
        want_schur = o["pc"] in ("cptr", "fieldsplit_cd")
        F = self.prob.residual(u)
        fnorm = float(np.linalg.norm(F))
        fnorm0 = fnorm
        nits, lits = 0, 0
        reason = 0
        hist = [fnorm]
        if not np.isfinite(fnorm):
            reason = SNES_DIVERGED_FNORM_NAN
        elif fnorm < o["snes_atol"]:
            reason = SNES_CONVERGED_FNORM_ABS
        while reason == 0:
            if nits >= o["snes_max_it"]:
                reason = SNES_DIVERGED_MAX_IT
                break
            out = self.prob.jacobian(u, want_schur=want_schur)
            J, Sm = out if want_schur else (out, None)
            dx, kits, kreason, _ = self.linear_solve(J, Sm, F)
            lits += kits
            if kreason < 0:
                reason = SNES_DIVERGED_LINEAR_SOLVE
                break
            u = u - dx                                    # basic line search, lambda = 1
            F = self.prob.residual(u)
            fnorm = float(np.linalg.norm(F))
            nits += 1
            hist.append(fnorm)
            snorm = float(np.linalg.norm(dx))
            xnorm = float(np.linalg.norm(u))
            if not np.isfinite(fnorm):
                reason = SNES_DIVERGED_FNORM_NAN
            elif fnorm < o["snes_atol"]:
                reason = SNES_CONVERGED_FNORM_ABS
            elif fnorm <= o["snes_rtol"] * fnorm0:
                reason = SNES_CONVERGED_FNORM_RELATIVE
            elif snorm < o["snes_stol"] * xnorm:
                reason = SNES_CONVERGED_SNORM_RELATIVE
        self.u = u
        self.last = dict(nits=nits, lits=lits, reason=reason, fnorm=fnorm, fnorm0=fnorm0, history=hist)
        return self.last
