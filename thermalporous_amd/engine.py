"""ctypes binding of libthermalporous_hip.so -- the compute engine behind SinglePhase/TwoPhase.solve().

This is the FFI stub a maintainer of the reference would add: the reference reaches
its native hot path through petsc4py/Firedrake at ``self.solver.solve()``
(/root/reference/thermalporous/thermalmodel.py:165); here the same call lands in ``tp_newton_solve``.
There is NO CPU fallback: without the HIP library or without a GPU the engine raises.

Slab layout: each rank owns planes [off2, off2+n2) along internal axis 2 and stores every cell array
with one halo plane per side (include/thermalporous_hip.h).
"""
import ctypes as C
import os

import numpy as np

_LIB = None
_LIBPATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libthermalporous_hip.so")


class EngineError(RuntimeError):
    pass


class tp_grid(C.Structure):
    _fields_ = [("n0", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32), ("gn2", C.c_int32), ("off2", C.c_int32),
                ("h", C.c_double*3), ("gaxis", C.c_int32), ("nphase", C.c_int32), ("rank", C.c_int32),
                ("nranks", C.c_int32)]


class tp_params(C.Structure):
    _names = ("ko", "kw", "kr", "c_v_w", "c_v_o", "c_r", "rho_r", "p_inj", "p_prod", "T_inj", "T_prod", "API",
              "p_ref", "g", "S_o", "U", "rate")
    _fields_ = [(k, C.c_double) for k in _names]


class tp_source(C.Structure):
    _fields_ = [("cell", C.c_int64), ("kind", C.c_int32), ("constant_rate", C.c_int32), ("wt", C.c_double),
                ("bhp", C.c_double), ("max_rate", C.c_double), ("WI", C.c_double)]


class tp_options(C.Structure):
    _fields_ = [("pc_kind", C.c_int32), ("decoup", C.c_int32), ("ksp_rtol", C.c_double), ("ksp_atol", C.c_double),
                ("ksp_max_it", C.c_int32), ("ksp_restart", C.c_int32), ("snes_rtol", C.c_double),
                ("snes_atol", C.c_double), ("snes_stol", C.c_double), ("snes_max_it", C.c_int32),
                ("amg_omega", C.c_double), ("amg_nu", C.c_int32), ("amg_min_cells", C.c_int32),
                ("ilu_t1", C.c_int32), ("ilu_t2", C.c_int32), ("ilu_t0", C.c_int32),
                ("amg_full_levels", C.c_int32), ("amg_coarse_pre", C.c_int32), ("amg_coarse_post", C.c_int32),
                ("amg_mid_skip", C.c_int32), ("amg_tail_post", C.c_int32), ("amg_single", C.c_int32), ("schur_a11", C.c_int32), ("amg_gather_cells", C.c_int32), ("amg_dom_tau", C.c_double),
                ("ilu_levels", C.c_int32), ("fs_additive", C.c_int32), ("ilu_whole", C.c_int32)]


class tp_solve_info(C.Structure):
    _fields_ = [("nits", C.c_int32), ("lits", C.c_int32), ("reason", C.c_int32), ("last_ksp_reason", C.c_int32),
                ("fnorm0", C.c_double), ("fnorm", C.c_double), ("vcycles", C.c_int32)]


# every symbol include/thermalporous_hip.h declares (tests check that the library exports them all)
API_SYMBOLS = (
    "tp_last_error", "tp_version", "tp_create", "tp_destroy", "tp_set_options", "tp_comm_unique_id", "tp_comm_init",
    "tp_local_group_create", "tp_local_group_destroy", "tp_comm_init_local",
    "tp_set_field", "tp_finalize_fields", "tp_set_sources", "tp_set_state", "tp_get_state", "tp_set_old_state",
    "tp_set_dt", "tp_get_old_state", "tp_restore_state", "tp_saturation_range", "tp_clamp_saturation",
    "tp_residual", "tp_jacobian", "tp_get_residual", "tp_export_jacobian", "tp_export_schur",
    "tp_well_rates", "tp_vec_create", "tp_vec_create_batch", "tp_vec_dot_batch", "tp_vec_axpy_batch", "tp_vec_norm2", "tp_set_ksp_monitor", "tp_vec_set", "tp_vec_get", "tp_vec_copy_residual", "tp_spmv", "tp_pc_setup",
    "tp_pc_apply", "tp_stage1_update", "tp_stage1_apply", "tp_ilu0_factor", "tp_ilu0_solve", "tp_amg_setup",
    "tp_amg_vcycle", "tp_schur_apply", "tp_fgmres", "tp_newton_solve", "tp_time_kernel", "tp_amg_info", "tp_amg_layout", "tp_amg_trunc",
)

DEFAULT_OPTS = dict(
    pc="cpr", decoup="No",
    ksp_rtol=1e-7, ksp_atol=1e-50, ksp_max_it=200, ksp_restart=200,
    snes_rtol=1e-8, snes_atol=1e-50, snes_stol=1e-8, snes_max_it=15,
    amg_omega=0.9,          # damped-Jacobi weight (round 3: 0.8 -> 0.9 buys 3 % fewer Krylov iterations on C4 at equal cycle cost, +4 % Newton steps/s
                            # over 80 time steps, measured twice; 0.88-0.9 is a plateau, 0.95 starts to fail solves, 1.0 loses 40 %; C1-C3 neutral)
    amg_min_cells=64, amg_nu=2, amg_full_levels=3, amg_coarse_pre=0, amg_coarse_post=1, amg_mid_skip=True, amg_tail_post=2, amg_single=False,
    amg_dom_tau=0.25,       # relaxation-only truncation of diagonally dominant AMG hierarchies (oracle/linalg.py:SemiAMG)
    # multi-GPU: AMG levels with more cells than this stay distributed over the slabs.  Cost model: a V(2,2)
    # level streams ~6 sweeps x 104 B per cell (5.5 TB/s on one GPU) and needs 6 halo exchanges when distributed; with N
    # slabs it saves (1 - 1/N) of its streaming time and pays 6 x t_exchange (~10 us per grouped RCCL send/recv): the
    # break-even is ~0.6 M cells, so C4's level 0 (1.12 M cells) is distributed, its level 1 (0.56 M) is gathered
    amg_gather_cells=600000,
    schur_a11=False,
    fs_additive=False,      # pc_fieldsplit_type additive on (p,T): pc_fieldsplit_diag (singlephase.py:371-375)
    schur_selfp=False,      # pc_fieldsplit_schur_precondition selfp (pc_fieldsplit_selfp, singlephase.py:322-330)
    ilu_tile=None,          # None: see default_ilu_tile (3-D: whole axis-0 lines x a balanced t1 x t2; 2-D: ~24 x 32 cells)
    ilu_levels=0,           # sub_1_sub_pc_factor_levels: 0 (block-ILU(0)) or 1 (block-ILU(1), pc_cprilu1_gmres)
    bjacobi_blocks=None,    # -sub_1_pc_bjacobi_blocks N: N blocks over the whole grid (tiles_for_blocks); overrides ilu_tile
    ilu_whole=False,        # one bjacobi block per rank: block-ILU(0) of the whole slab (= bjacobi_blocks 1 on one GPU, PETSc's
                            # default bjacobi on several); ilu_tile is then only the unit of the diagonal-by-diagonal sweep
)

def default_ilu_tile(n, nslabs=1, ncu=256):
    """bjacobi tile (t0, t1, t2) for a grid of internal extents n = (n0, n1, n2) cut into `nslabs` slabs along axis 2.
    2-D: 32 columns (measured on C3 60x220: 64-wide tiles cost 123 wavefront steps for 60 cells of depth, 32-wide ones
    91 steps and +0.5 % Krylov iterations) x pieces of ~24 cells of the axis-0 lines (below).  3-D: whole axis-0 lines
    (the thin, strongly coupled direction; the sweeps are bandwidth bound there and shorter tiles only add fill/drain steps) x the t1 x t2 (32..64 columns, each side
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


def whole_ilu_tile(n, nslabs=1):
    """Sweep unit (t0, t1, t2) of the whole-slab ILU(0) (``ilu_whole``): the tiles no longer cut couplings, they are swept
    one tile-diagonal T0 + T1 + T2 = d per launch.  3-D: the balanced t1 x t2 of default_ilu_tile, and the axis-0 lines cut
    into pieces of ~16 cells -- a diagonal of the 3-D tile grid holds up to nt0*nt1 tiles instead of nt1, and a launch
    lasts t0 + t1 + t2 - 2 wavefront steps instead of n0 + t1 + t2 - 2 (C4: 39 launches of ~29 steps per direction instead
    of 34 of 98).  2-D: the 24 x 32 tiles of the default."""
    n0, n1, n2 = (int(v) for v in n)
    t = default_ilu_tile(n, nslabs=nslabs)
    if n2 == 1:
        return t
    return (-(-n0//max(1, -(-n0//16))), t[1], t[2])


def tiles_for_blocks(n, nblocks, max_cols=64):
    """Tile (t0, t1, t2) that cuts the grid n = (n0, n1, n2) into exactly `nblocks` boxes = bjacobi blocks
    (``-sub_1_pc_bjacobi_blocks``, /root/reference/tests/test_homo_wells.py:112,125).  PETSc's blocks are contiguous row
    ranges of its field-major DMPlex ordering, which has no counterpart here; the build's blocks are boxes of whole
    cells, whole axis-0 lines when possible.  A GPU tile is swept by one wavefront: at most `max_cols` = 64 columns
    (t1*t2); None lifts the limit (CPU oracle).  Raises NotImplementedError when no such tiling exists."""
    n0, n1, n2 = (int(v) for v in n)
    nblocks = int(nblocks)
    if nblocks < 1:
        raise ValueError("bjacobi_blocks must be >= 1")
    best = None
    if nblocks == 1 and max_cols is not None and n1*n2 > max_cols:
        raise NotImplementedError("one block over a grid of more than %d columns is not a single tile: the GPU engine realises "
                                  "it as ilu_whole (whole-slab ILU(0) swept tile-diagonal by tile-diagonal)" % max_cols)
    for k2 in range(1, min(n2, nblocks) + 1):
        if nblocks % k2:
            continue
        rem = nblocks//k2
        for k1 in range(1, min(n1, rem) + 1):
            if rem % k1:
                continue
            k0 = rem//k1
            if k0 > n0:
                continue
            t = [-(-n0//k0), -(-n1//k1), -(-n2//k2)]
            if (-(-n0//t[0]), -(-n1//t[1]), -(-n2//t[2])) != (k0, k1, k2):
                continue
            if max_cols is not None and t[1]*t[2] > max_cols:
                continue
            score = (k0 != 1, t[0]*t[1] + t[1]*t[2] + t[0]*t[2])      # whole lines first, then the most compact box
            if best is None or score < best[0]:
                best = (score, tuple(t))
    if best is None:
        raise NotImplementedError(
            "sub_1_pc_bjacobi_blocks = %d cannot be realised on a %dx%dx%d grid with bjacobi tiles of at most %s columns "
            "(one wavefront sweeps a tile; whole-grid ILU(0) needs n1*n2 <= 64 here).  Leave the key out for the "
            "engine's default tiles, or pass ilu_tile." % (nblocks, n0, n1, n2, max_cols))
    return best[1]


_PC = {"cpr": 0, "cptr": 1, "fieldsplit_cd": 2, "cptramg": 3, "bilu": 4}
_DECOUP = {"No": 0, "QI": 1, "TI": 2, "QI_temp": 3, "TI_temp": 4}


def load_library(path=None):
    """dlopen the in-tree HIP library; fail loudly if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or _LIBPATH
    if not os.path.exists(p):
        raise EngineError("libthermalporous_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    lib = C.CDLL(p)
    lib.tp_last_error.restype = C.c_char_p
    for name in API_SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        if name != "tp_last_error":
            fn.restype = C.c_int
    if path is None:
        _LIB = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def slab_range(gn2, rank, nranks):
    """Planes [lo, hi) of internal axis 2 owned by `rank` (as even as possible, low ranks get the extras)."""
    base, rem = divmod(gn2, nranks)
    lo = rank*base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipEngine:
    """Same interface as oracle.engine.OracleEngine; all arithmetic on the GPU."""

    def __init__(self, spec, opts=None, rank=0, nranks=1, device=None, comm_bootstrap=None, local_group=None):
        self.lib = load_library()
        self.spec = spec
        self.opts = dict(DEFAULT_OPTS)
        self.opts.update(opts or {})
        if self.opts.get("bjacobi_blocks") is not None:
            # PETSc counts blocks over ALL ranks and needs at least one per rank: N blocks on N slabs = one per rank
            if int(self.opts["bjacobi_blocks"]) == int(nranks):
                self.opts["ilu_whole"] = True
            elif int(nranks) > 1:
                raise EngineError("bjacobi_blocks counts blocks over the whole grid: on multi-slab runs only one block per "
                                  "rank (bjacobi_blocks = number of slabs, or ilu_whole) or an explicit ilu_tile")
            else:
                self.opts["ilu_tile"] = tiles_for_blocks(spec["n"], self.opts["bjacobi_blocks"], max_cols=64)
        if self.opts["ilu_tile"] is None:
            self.opts["ilu_tile"] = (whole_ilu_tile if self.opts.get("ilu_whole") else default_ilu_tile)(spec["n"], nslabs=int(nranks))
        self.nph = int(spec["nphase"])
        self.b = self.nph + 1
        n0, n1, gn2 = (int(v) for v in spec["n"])
        self.rank, self.nranks = int(rank), int(nranks)
        if self.nranks > 1 and gn2 < self.nranks:
            raise EngineError("more slabs than planes along the slab axis")
        self.lo, self.hi = slab_range(gn2, self.rank, self.nranks)
        self.n = (n0, n1, self.hi - self.lo)
        self.gn = (n0, n1, gn2)
        self.np_ = n0*n1
        self.ntot = self.np_*(self.n[2] + 2)
        g = tp_grid(n0, n1, self.n[2], gn2, self.lo, (C.c_double*3)(*[float(h) for h in spec["h"]]),
                    int(spec["gaxis"]), self.nph, self.rank, self.nranks)
        prm = tp_params(*[float(spec["prm"][k]) for k in tp_params._names])
        self._opt = self._make_options(self.opts)
        self.ctx = C.c_void_p()
        if device is None:
            device = int(os.environ.get("TP_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if (self.nranks > 1 and local_group is None) else 0
        self._ck(self.lib.tp_create(C.byref(g), C.byref(prm), C.byref(self._opt), int(device), C.byref(self.ctx)))
        if self.nranks > 1 and local_group is not None:
            # N engines in N threads of this process sharing one GPU (validation of the slab algorithm)
            self._ck(self.lib.tp_comm_init_local(self.ctx, local_group))
        elif self.nranks > 1 or comm_bootstrap is not None:
            # (nranks == 1 with an explicit bootstrap: a one-rank RCCL communicator, used by the tests to run the
            # library's RCCL calls for real on a one-GPU box)
            if comm_bootstrap is None:
                raise EngineError("multi-slab engine needs comm_bootstrap(make_id) -> 128-byte id")
            ident = comm_bootstrap(self._unique_id)
            self._ck(self.lib.tp_comm_init(self.ctx, C.c_char_p(bytes(ident))))
        # fields (slab + halo planes; halo of a physical boundary replicates the boundary plane)
        for name, arr in (("phi", spec["phi"]), ("kT", spec["kT"]), ("K0", spec["K"][0]), ("K1", spec["K"][1]),
                          ("K2", spec["K"][2])):
            a = self._with_halo(np.asarray(arr, dtype=float))
            self._ck(self.lib.tp_set_field(self.ctx, name.encode(), _dptr(a), C.c_int64(a.size)))
        self._ck(self.lib.tp_finalize_fields(self.ctx))
        self._set_sources(spec.get("sources"))
        self.last = {}
        self._vec_ids = {}

    # ---- plumbing ---------------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise EngineError(self.lib.tp_last_error().decode())

    def _unique_id(self):
        buf = C.create_string_buffer(128)
        self._ck(self.lib.tp_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def _make_options(o):
        t = o["ilu_tile"]
        return tp_options(_PC[o["pc"]], _DECOUP[o["decoup"]], o["ksp_rtol"], o["ksp_atol"], o["ksp_max_it"],
                          o["ksp_restart"], o["snes_rtol"], o["snes_atol"], o["snes_stol"], o["snes_max_it"],
                          o["amg_omega"], o["amg_nu"], o["amg_min_cells"], int(min(t[1], 64)), int(min(t[2], 64)),
                          0 if t[0] >= (1 << 30) else int(t[0]), int(o["amg_full_levels"]), int(o["amg_coarse_pre"]),
                          int(o["amg_coarse_post"]), int(bool(o["amg_mid_skip"])), int(o["amg_tail_post"]), int(bool(o["amg_single"])), 2 if o.get("schur_selfp") else int(bool(o["schur_a11"])),
                          int(o["amg_gather_cells"]), float(o.get("amg_dom_tau", 0.0)), int(o.get("ilu_levels", 0)), int(bool(o.get("fs_additive", False))),
                          int(bool(o.get("ilu_whole", False))))

    def set_options(self, **kw):
        self.opts.update(kw)
        self._opt = self._make_options(self.opts)
        self._ck(self.lib.tp_set_options(self.ctx, C.byref(self._opt)))

    def _with_halo(self, a):
        """Global internal array (gn2, n1, n0) -> this slab with halo planes, flat, C-contiguous."""
        a = a.reshape(self.gn[2], self.gn[1], self.gn[0])
        lo, hi = self.lo, self.hi
        idx = np.clip(np.arange(lo - 1, hi + 1), 0, self.gn[2] - 1)
        return np.ascontiguousarray(a[idx]).reshape(-1)

    def _fields_with_halo(self, u):
        u = np.asarray(u, dtype=float).reshape(self.b, self.gn[2], self.gn[1], self.gn[0])
        return np.ascontiguousarray(np.stack([self._with_halo(u[f]) for f in range(self.b)])).reshape(-1)

    def _strip_halo(self, flat, nf):
        a = np.asarray(flat).reshape(nf, self.n[2] + 2, self.n[1], self.n[0])
        return a[:, 1:-1]

    def _set_sources(self, src):
        if not src or len(src["cell"]) == 0:
            self._ck(self.lib.tp_set_sources(self.ctx, 0, None))
            self.src_index = np.zeros(0, dtype=int)
            return
        cells = np.asarray(src["cell"], dtype=np.int64)
        plane = cells // self.np_
        mine = np.nonzero((plane >= self.lo) & (plane < self.hi))[0]
        self.src_index = mine                      # positions (in the global entry list) of my entries
        arr = (tp_source*max(1, len(mine)))()
        for k, i in enumerate(mine):
            local = int(cells[i] - self.lo*self.np_ + self.np_)
            arr[k] = tp_source(local, int(src["kind"][i]), int(src["const"][i]), float(src["wt"][i]),
                               float(src["bhp"][i]), float(src["max_rate"][i]), float(src["WI"][i]))
        self._ck(self.lib.tp_set_sources(self.ctx, len(mine), arr))
        # the library sorts entries by cell (stable): reproduce the permutation for rate read-back
        self._src_order = np.argsort(cells[mine], kind="stable")

    # ---- engine interface ---------------------------------------------------------------------------
    def set_state(self, u):
        a = self._fields_with_halo(u)
        self._ck(self.lib.tp_set_state(self.ctx, _dptr(a)))

    def get_state(self):
        """This rank's owned part of the state, shape (b, n2_local, n1, n0)."""
        out = np.empty(self.b*self.ntot)
        self._ck(self.lib.tp_get_state(self.ctx, _dptr(out)))
        return self._strip_halo(out, self.b).copy()

    def set_old(self, u=None):
        if u is None:
            self._ck(self.lib.tp_set_old_state(self.ctx, None))
        else:
            a = self._fields_with_halo(u)
            self._ck(self.lib.tp_set_old_state(self.ctx, _dptr(a)))

    def set_dt(self, dt):
        self._ck(self.lib.tp_set_dt(self.ctx, C.c_double(float(dt))))

    def get_old_state(self):
        out = np.empty(self.b*self.ntot)
        self._ck(self.lib.tp_get_old_state(self.ctx, _dptr(out)))
        return self._strip_halo(out, self.b).copy()

    def restore_state(self):
        self._ck(self.lib.tp_restore_state(self.ctx))

    def saturation_range(self):
        lo, hi = C.c_double(), C.c_double()
        self._ck(self.lib.tp_saturation_range(self.ctx, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def clamp_saturation(self):
        self._ck(self.lib.tp_clamp_saturation(self.ctx))

    def residual(self, u=None):
        if u is not None:
            self.set_state(u)
        nrm = C.c_double()
        self._ck(self.lib.tp_residual(self.ctx, C.byref(nrm)))
        out = np.empty(self.b*self.ntot)
        self._ck(self.lib.tp_get_residual(self.ctx, _dptr(out)))
        self.last_fnorm = nrm.value
        return self._strip_halo(out, self.b).copy()

    def jacobian(self, u=None, want_schur=False):
        if u is not None:
            self.set_state(u)
        if want_schur and self.opts["pc"] not in ("cptr", "fieldsplit_cd"):
            raise EngineError("S~ is assembled only for pc='cptr' / 'fieldsplit_cd'")
        self._ck(self.lib.tp_jacobian(self.ctx))
        b = self.b
        out = np.empty(7*b*b*self.ntot)
        self._ck(self.lib.tp_export_jacobian(self.ctx, _dptr(out)))
        J = self._strip_halo(out, 7*b*b).reshape(7, b, b, self.n[2], self.n[1], self.n[0]).copy()
        if want_schur:
            s = np.empty(7*self.ntot)
            self._ck(self.lib.tp_export_schur(self.ctx, _dptr(s)))
            return J, self._strip_halo(s, 7).copy()
        return J

    def well_rates(self):
        n = len(self.src_index)
        if n == 0:
            return {}
        r, w, o = (np.zeros(n) for _ in range(3))
        self._ck(self.lib.tp_well_rates(self.ctx, _dptr(r), _dptr(w), _dptr(o)))
        inv = np.empty(n, dtype=int)
        inv[self._src_order] = np.arange(n)
        return {"rate": r[inv], "water_rate": w[inv], "oil_rate": o[inv]}

    # device vectors for the PC plug-in API / tests
    def vec(self, name):
        if name not in self._vec_ids:
            i = C.c_int32()
            self._ck(self.lib.tp_vec_create(self.ctx, C.byref(i)))
            self._vec_ids[name] = i.value
        return self._vec_ids[name]

    def vec_set(self, name, x):
        a = self._fields_with_halo(x)
        self._ck(self.lib.tp_vec_set(self.ctx, self.vec(name), _dptr(a)))

    def vec_get(self, name):
        out = np.empty(self.b*self.ntot)
        self._ck(self.lib.tp_vec_get(self.ctx, self.vec(name), _dptr(out)))
        return self._strip_halo(out, self.b).copy()

    def vec_batch(self, prefix, n):
        """n vectors '<prefix>0'..'<prefix>{n-1}' in one allocation (a Krylov basis): VecMDot/VecMAXPY run in one pass."""
        if prefix + "0" not in self._vec_ids:
            i = C.c_int32()
            self._ck(self.lib.tp_vec_create_batch(self.ctx, int(n), C.byref(i)))
            for k in range(n):
                self._vec_ids[prefix + str(k)] = i.value + k
        return self._vec_ids[prefix + "0"]

    def dot_batch(self, prefix, n, w):
        out = np.zeros(n)
        self._ck(self.lib.tp_vec_dot_batch(self.ctx, self.vec_batch(prefix, n), int(n), self.vec(w), _dptr(out)))
        return out

    def axpy_batch(self, prefix, n, coef, w):
        coef = np.ascontiguousarray(coef, dtype=float)
        self._ck(self.lib.tp_vec_axpy_batch(self.ctx, self.vec_batch(prefix, n), int(n), _dptr(coef), self.vec(w)))

    def norm2(self, x):
        out = C.c_double()
        self._ck(self.lib.tp_vec_norm2(self.ctx, self.vec(x), C.byref(out)))
        return out.value

    def set_ksp_monitor(self, fn):
        """fn(its, rnorm, field_norms) at every FGMRES iteration (the reference's ksp_monitor_residuals monitor,
        thermalmodel.py:44-74); None removes it."""
        proto = C.CFUNCTYPE(None, C.c_int32, C.c_double, C.POINTER(C.c_double), C.c_int32, C.c_void_p)
        if fn is None:
            self._monitor_cb = None
            self._ck(self.lib.tp_set_ksp_monitor(self.ctx, C.cast(None, proto), None))
            return
        self._monitor_cb = proto(lambda its, rn, fnp, nf, user: fn(int(its), float(rn), [fnp[i] for i in range(nf)]))
        self._ck(self.lib.tp_set_ksp_monitor(self.ctx, self._monitor_cb, None))

    def spmv(self, x, y):
        self._ck(self.lib.tp_spmv(self.ctx, self.vec(x), self.vec(y)))

    def pc_setup(self):
        self._ck(self.lib.tp_pc_setup(self.ctx))

    def pc_apply(self, x, y):
        self._ck(self.lib.tp_pc_apply(self.ctx, self.vec(x), self.vec(y)))

    def stage1_apply(self, x, y):
        self._ck(self.lib.tp_stage1_apply(self.ctx, self.vec(x), self.vec(y)))

    def ilu_solve(self, x, y):
        self._ck(self.lib.tp_ilu0_solve(self.ctx, self.vec(x), self.vec(y)))

    def schur_apply(self, x, y):
        self._ck(self.lib.tp_schur_apply(self.ctx, self.vec(x), self.vec(y)))

    def vec_axpby(self, out, a, x, b, y):
        """out = a*x + b*y through the host (test/plug-in convenience, not on the hot path)."""
        self.vec_set(out, a*self._full(self.vec_get(x)) + b*self._full(self.vec_get(y)))

    def _full(self, owned):
        """Owned slab part -> global-shaped array (single-slab engines only)."""
        if self.nranks != 1:
            raise EngineError("host-side vector algebra is only available on single-slab engines")
        return owned

    def amg_vcycle(self, which, b, fb, x, fx):
        self._ck(self.lib.tp_amg_vcycle(self.ctx, which, fb, self.vec(b), fx, self.vec(x)))

    def fgmres(self, b, x):
        its, reason, rn = C.c_int32(), C.c_int32(), C.c_double()
        self._ck(self.lib.tp_fgmres(self.ctx, self.vec(b), self.vec(x), C.byref(its), C.byref(reason), C.byref(rn)))
        return its.value, reason.value, rn.value

    def copy_residual_to(self, name):
        self._ck(self.lib.tp_vec_copy_residual(self.ctx, self.vec(name)))

    def time_kernel(self, which, reps):
        ms = C.c_double()
        self._ck(self.lib.tp_time_kernel(self.ctx, which, reps, C.byref(ms)))
        return ms.value

    def amg_info(self, which=0):
        nl, oc = C.c_int32(), C.c_double()
        self._ck(self.lib.tp_amg_info(self.ctx, which, C.byref(nl), C.byref(oc)))
        return nl.value, oc.value

    def amg_trunc(self, which=0):
        """(level at which hierarchy `which` ends with relaxation only, -1 = full V-cycle; level-0 dominance ratio)."""
        lv, r0 = C.c_int32(), C.c_double()
        self._ck(self.lib.tp_amg_trunc(self.ctx, which, C.byref(lv), C.byref(r0)))
        return lv.value, r0.value

    def amg_layout(self, which=0):
        """(number of slab-distributed top levels, coarsening axis of every level)."""
        nd, na = C.c_int32(), C.c_int32()
        axes = (C.c_int32*64)()
        self._ck(self.lib.tp_amg_layout(self.ctx, which, C.byref(nd), axes, 64, C.byref(na)))
        return nd.value, [axes[i] for i in range(min(na.value, 64))]

    def newton_solve(self):
        info = tp_solve_info()
        self._ck(self.lib.tp_newton_solve(self.ctx, C.byref(info)))
        self.last = dict(nits=info.nits, lits=info.lits, reason=info.reason, fnorm=info.fnorm, fnorm0=info.fnorm0,
                         ksp_reason=info.last_ksp_reason, vcycles=info.vcycles)
        return self.last

    def close(self):
        if self.ctx:
            self.lib.tp_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
