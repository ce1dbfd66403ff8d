"""GPU parity tests: every stage of the HIP hot path against the CPU oracle on the same seeded inputs.

Tolerances (float64 everywhere; differences come only from FMA contraction / summation order):
  residual, Jacobian, S~ entries   rel <= 1e-11 of the largest entry of the same field/plane
  SpMV, ILU solve, AMG V-cycle     rel <= 1e-10 in the 2-norm
  FGMRES / Newton                  same iteration counts (+-1 Krylov), converged state rel <= 1e-8
The reference itself (Firedrake/PETSc) cannot run here: parity vs the reference is unpinned, see
oracle/__init__.py."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.abs(a - b).max()/max(np.abs(b).max(), 1e-300)


def rel2(a, b):
    return np.linalg.norm((a - b).ravel())/max(np.linalg.norm(b.ravel()), 1e-300)


def make(builder, opts, **kw):
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = builder(**kw)
    o = OracleEngine(spec, opts)
    h = HipEngine(spec, opts)
    return spec, u0, o, h


CASES = [
    ("c1_1ph_2d", cases.c1_homogeneous, dict(N=12, nphase=1), dict(pc="cpr", ilu_tile=(1 << 30, 64, 1))),
    ("c3_2ph_2d", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=2), dict(pc="cptr", ilu_tile=(1 << 30, 64, 1))),
    ("c2_1ph_2d", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=1), dict(pc="cpr", decoup="QI", ilu_tile=(1 << 30, 64, 1))),
    ("c4_2ph_3d", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=2), dict(pc="cptr")),
    ("c4_1ph_3d", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="cpr", decoup="TI")),
    # a plane of fewer than 8 cells (n0*n1 = 6): the multi-wave ILU sweep must fall back from its 8-value block transfers
    ("c4_2ph_3d_tinyplane", cases.c4_spe10_3d, dict(Nx=3, Ny=14, Nz=2, nphase=2), dict(pc="cptr")),
    ("c4_2ph_3d_cprQI", cases.c4_spe10_3d, dict(Nx=9, Ny=10, Nz=5, nphase=2), dict(pc="cpr", decoup="QI")),
    ("c4_2ph_3d_tiles", cases.c4_spe10_3d, dict(Nx=11, Ny=13, Nz=17, nphase=2), dict(pc="cptr", ilu_tile=(5, 4, 7))),
    ("c4_2ph_3d_fp32amg", cases.c4_spe10_3d, dict(Nx=9, Ny=14, Nz=8, nphase=2), dict(pc="cptr", amg_single=True)),
    ("c4_2ph_3d_v22", cases.c4_spe10_3d, dict(Nx=9, Ny=14, Nz=8, nphase=2), dict(pc="cptr", amg_full_levels=99)),
    ("c4_2ph_3d_v02", cases.c4_spe10_3d, dict(Nx=9, Ny=14, Nz=8, nphase=2), dict(pc="cptr", amg_full_levels=2, amg_coarse_post=2)),
    ("c4_2ph_3d_v11c", cases.c4_spe10_3d, dict(Nx=9, Ny=14, Nz=8, nphase=2), dict(pc="cptr", amg_full_levels=1, amg_coarse_pre=1, amg_coarse_post=1)),
    # pure transfer levels (amg_mid_skip): 4608 -> 2304 (smoothed) -> 1152 (transfer only) -> 576 ... with one full level
    ("c4_2ph_3d_midskip", cases.c4_spe10_3d, dict(Nx=16, Ny=18, Nz=16, nphase=2), dict(pc="cptr", amg_full_levels=1)),
    ("c4_2ph_3d_midskip2", cases.c4_spe10_3d, dict(Nx=20, Ny=26, Nz=18, nphase=2), dict(pc="cptr", amg_full_levels=1, amg_single=True)),
    ("c4_2ph_3d_nomidskip", cases.c4_spe10_3d, dict(Nx=16, Ny=18, Nz=16, nphase=2), dict(pc="cptr", amg_full_levels=1, amg_mid_skip=False)),
    # QI_temp / TI_temp: temperature AND saturation decoupled from the pressure (preconditioners.py:714-783, 810-873)
    ("c4_2ph_3d_cprQItemp", cases.c4_spe10_3d, dict(Nx=9, Ny=10, Nz=5, nphase=2), dict(pc="cpr", decoup="QI_temp")),
    ("c4_2ph_3d_cprTItemp", cases.c4_spe10_3d, dict(Nx=7, Ny=8, Nz=6, nphase=2), dict(pc="cpr", decoup="TI_temp")),
    # Schur complement preconditioned by A_11 instead of S~ (pc_cptr_a11 twophase.py:598-616, pc_fieldsplit_a11)
    ("c4_2ph_3d_cptr_a11", cases.c4_spe10_3d, dict(Nx=7, Ny=8, Nz=6, nphase=2), dict(pc="cptr", schur_a11=True)),
    ("c4_2ph_3d_cptrQI_a11", cases.c4_spe10_3d, dict(Nx=7, Ny=8, Nz=6, nphase=2), dict(pc="cptr", decoup="QI", schur_a11=True)),
    ("c4_1ph_3d_fs_a11", cases.c4_spe10_3d, dict(Nx=7, Ny=8, Nz=6, nphase=1), dict(pc="fieldsplit_cd", schur_a11=True)),
    # degenerate boxes: one cell wide in one or two directions, a 2x2x2 cube, a single line, more columns than a wave has lanes
    ("deg_1x5x7_cptr", cases.c4_spe10_3d, dict(Nx=1, Ny=5, Nz=7, nphase=2), dict(pc="cptr")),
    ("deg_2x2x2_cptr", cases.c4_spe10_3d, dict(Nx=2, Ny=2, Nz=2, nphase=2), dict(pc="cptr")),
    ("deg_line_cptr", cases.c4_spe10_3d, dict(Nx=1, Ny=1, Nz=9, nphase=2), dict(pc="cptr")),
    ("deg_3x4x1_cptr", cases.c4_spe10_3d, dict(Nx=3, Ny=4, Nz=1, nphase=2), dict(pc="cptr")),
    ("deg_70x3x2_cpr", cases.c4_spe10_3d, dict(Nx=70, Ny=3, Nz=2, nphase=1), dict(pc="cpr")),
    ("deg_2x2x2_ilu1", cases.c4_spe10_3d, dict(Nx=2, Ny=2, Nz=2, nphase=2), dict(pc="cpr", ilu_levels=1)),
    ("deg_1x5x7_ilu1", cases.c4_spe10_3d, dict(Nx=1, Ny=5, Nz=7, nphase=2), dict(pc="cpr", ilu_levels=1)),
    ("deg_1x5x7_whole", cases.c4_spe10_3d, dict(Nx=1, Ny=5, Nz=7, nphase=2), dict(pc="cptr", ilu_whole=True, ilu_tile=(1 << 30, 2, 3))),
    ("deg_2x3x2_cptramg", cases.c4_spe10_3d, dict(Nx=2, Ny=3, Nz=2, nphase=2), dict(pc="cptramg", decoup="QI")),
    # pc_cptramg[_QI|_TI] (twophase.py:552-566): one system-AMG V-cycle on the 2x2-block (p,T) operator as stage 1
    ("c4_2ph_3d_cptramg", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=2), dict(pc="cptramg")),
    ("c4_2ph_3d_cptramg_QI", cases.c4_spe10_3d, dict(Nx=9, Ny=14, Nz=8, nphase=2), dict(pc="cptramg", decoup="QI")),
    ("c4_2ph_3d_cptramg_TI", cases.c4_spe10_3d, dict(Nx=16, Ny=18, Nz=16, nphase=2), dict(pc="cptramg", decoup="TI", amg_full_levels=1)),
    ("c3_2ph_2d_cptramg", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=2), dict(pc="cptramg", decoup="QI", ilu_tile=(1 << 30, 64, 1))),
    # pc_fieldsplit_selfp (singlephase.py:322-330): Sp = A11 - A10 diag(A00)^-1 A01 preconditions the Schur complement
    ("c4_1ph_3d_selfp", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="fieldsplit_cd", schur_selfp=True)),
    ("c2_1ph_2d_selfp", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=1), dict(pc="fieldsplit_cd", schur_selfp=True)),
    ("c1_1ph_2d_selfp", cases.c1_homogeneous, dict(N=12, nphase=1), dict(pc="fieldsplit_cd", schur_selfp=True)),
    # pc_fieldsplit_diag (singlephase.py:371-375): additive fieldsplit, one V-cycle on A_pp and one on A_TT
    ("c4_1ph_3d_fsdiag", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="fieldsplit_cd", schur_a11=True, fs_additive=True)),
    ("c2_1ph_2d_fsdiag", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=1), dict(pc="fieldsplit_cd", schur_a11=True, fs_additive=True)),
    # block-ILU(1) second stage (pc_cprilu1_gmres, twophase.py:653-668): partial tiles, whole-line tiles, 2-D, 2x2 blocks
    ("c4_2ph_3d_ilu1_tiles", cases.c4_spe10_3d, dict(Nx=11, Ny=13, Nz=17, nphase=2), dict(pc="cpr", ilu_levels=1, ilu_tile=(5, 4, 7))),
    ("c4_2ph_3d_cptr_ilu1", cases.c4_spe10_3d, dict(Nx=9, Ny=14, Nz=8, nphase=2), dict(pc="cptr", ilu_levels=1)),
    ("c3_2ph_2d_ilu1", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=2), dict(pc="cpr", ilu_levels=1, ilu_tile=(1 << 30, 64, 1))),
    ("c2_1ph_2d_ilu1", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=1), dict(pc="cpr", decoup="QI", ilu_levels=1, ilu_tile=(6, 5, 1))),
    ("c4_1ph_3d_ilu1", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="cpr", decoup="TI", ilu_levels=1, ilu_tile=(1 << 30, 8, 8))),
    # ONE bjacobi block = block-ILU(0) of the whole grid (sub_1_pc_bjacobi_blocks: 1, tests/test_homo_wells.py:112; pc_cptr_a11
    # twophase.py:612): tiles keep their couplings and are swept tile-diagonal by tile-diagonal; partial tiles on every axis,
    # axis-0 cuts, 2-D, 2x2 blocks, per-step and block vector transfers (the oracle's ILU has one tile = the grid)
    ("c4_2ph_3d_whole", cases.c4_spe10_3d, dict(Nx=11, Ny=13, Nz=17, nphase=2), dict(pc="cptr", ilu_whole=True, ilu_tile=(5, 4, 7))),
    ("c4_2ph_3d_whole_dflt", cases.c4_spe10_3d, dict(Nx=9, Ny=22, Nz=37, nphase=2), dict(pc="cptr", bjacobi_blocks=1)),
    ("c4_1ph_3d_whole", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="cpr", decoup="QI", ilu_whole=True, ilu_tile=(1 << 30, 3, 4))),
    ("c3_2ph_2d_whole", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=2), dict(pc="cptr", ilu_whole=True, ilu_tile=(6, 5, 1))),
    ("c2_1ph_2d_whole", cases.c3_spe10_2d, dict(Nx=30, Ny=70, nphase=1), dict(pc="cpr", bjacobi_blocks=1)),
    ("c4_2ph_3d_cptr_a11_whole", cases.c4_spe10_3d, dict(Nx=7, Ny=8, Nz=6, nphase=2), dict(pc="cptr", schur_a11=True, bjacobi_blocks=1)),
    # single-phase block preconditioner pc_fieldsplit_cd (singlephase.py:309-319): ConvDiffSchurPC operator
    ("c4_1ph_3d_fscd", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="fieldsplit_cd")),
    ("c2_1ph_2d_fscd", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=1), dict(pc="fieldsplit_cd")),
]


@pytest.mark.parametrize("name,builder,kw,opts", CASES, ids=[c[0] for c in CASES])
def test_assembly_parity(name, builder, kw, opts):
    spec, u0, o, h = make(builder, opts, **kw)
    u = cases.perturbed_state(spec, seed=3)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(8640.0)
        e.set_state(u)
    Ro = o.residual()
    Rh = h.residual()
    for f in range(o.b):
        assert relmax(Rh[f], Ro[f]) < 1e-11, (name, "residual field", f)
    schur = opts["pc"] in ("cptr", "fieldsplit_cd")
    out_o = o.jacobian(want_schur=schur)
    out_h = h.jacobian(want_schur=schur)
    Jo, Jh = (out_o[0], out_h[0]) if schur else (out_o, out_h)
    b = o.b
    for r in range(b):
        for c in range(b):
            scale = np.abs(Jo[:, r, c]).max()
            if scale == 0.0:
                assert np.abs(Jh[:, r, c]).max() == 0.0
                continue
            assert np.abs(Jh[:, r, c] - Jo[:, r, c]).max()/scale < 1e-11, (name, "J block", r, c)
    if schur:
        assert relmax(out_h[1], out_o[1]) < 1e-11
    # well rates
    ro, rh = o.well_rates(), h.well_rates()
    for k in ro:
        assert np.allclose(rh[k], ro[k], rtol=1e-12, atol=1e-30), k
    h.close()


@pytest.mark.parametrize("name,builder,kw,opts", CASES, ids=[c[0] for c in CASES])
def test_linear_stages_parity(name, builder, kw, opts):
    import oracle.linalg as la
    spec, u0, o, h = make(builder, opts, **kw)
    u = cases.perturbed_state(spec, seed=5, amp=0.3)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(8640.0)
        e.set_state(u)
    schur = opts["pc"] in ("cptr", "fieldsplit_cd")
    out = o.jacobian(want_schur=schur)
    J, Sm = out if schur else (out, None)
    h.jacobian()
    rng = np.random.default_rng(11)
    x = rng.standard_normal(J.shape[1:2] + J.shape[3:])
    # MatMult
    h.vec_set("x", x)
    h.spmv("x", "y")
    assert rel2(h.vec_get("y"), la.spmv_block(J, x)) < 1e-12
    # PC set-up, then each stage
    o.pc.setup(J, Sm)
    h.pc_setup()
    h.ilu_solve("x", "y")
    assert rel2(h.vec_get("y"), o.pc.ilu.solve(x)) < 1e-10
    if opts["pc"] == "cptramg":
        h.amg_vcycle(2, "x", 0, "y", 0)          # the system V-cycle on fields (p,T)
        assert rel2(h.vec_get("y")[:2], o.pc.amg_pT.vcycle(x[:2])) < 1e-10
        # TI: the column sums cancel to ~1e-9 of their terms, so the summation order shows in d = D_0s/D_ss
        stol = 1e-7 if opts.get("decoup") == "TI" else 1e-10
        h.stage1_apply("x", "y")
        assert rel2(h.vec_get("y"), o.pc.stage1(x)) < stol
        h.pc_apply("x", "y")
        assert rel2(h.vec_get("y"), o.pc.apply(x)) < stol
        F = o.residual()
        h.residual()
        h.copy_residual_to("b")
        its_h, reason_h, _ = h.fgmres("b", "d")
        d_o, its_o, reason_o, _ = la.fgmres(lambda v: la.spmv_block(J, v), o.pc.apply, F, rtol=o.opts["ksp_rtol"],
                                            maxit=o.opts["ksp_max_it"], restart=o.opts["ksp_restart"])
        assert reason_h == reason_o == 2 and abs(its_h - its_o) <= 1, (its_h, its_o)
        assert rel2(h.vec_get("d"), d_o) < 1e-6
        h.close()
        return
    h.amg_vcycle(0, "x", 0, "y", 0)
    # amg_single: both sides round the stored operators to fp32 identically; the double-precision
    # intermediates differ by FMA contraction before that rounding, so a few entries round differently
    vtol = 1e-6 if opts.get("amg_single") else 1e-10
    if opts.get("decoup") == "TI_temp":
        vtol = 1e-6      # column sums cancel to ~1e-9 of their terms and pass through a 2x2 inverse: summation order shows
    assert rel2(h.vec_get("y")[0], o.pc.amg_p.vcycle(x[0])) < vtol
    if schur:
        # same relaxation-only truncation decision (amg_dom_tau) on both sides
        assert h.amg_trunc(1)[0] == (-1 if o.pc.amg_T.trunc is None else o.pc.amg_T.trunc)
        assert h.amg_trunc(0)[0] == (-1 if o.pc.amg_p.trunc is None else o.pc.amg_p.trunc)
        h.amg_vcycle(1, "x", 1, "y", 1)
        assert rel2(h.vec_get("y")[1], o.pc.amg_T.vcycle(x[1])) < vtol
    h.stage1_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.stage1(x)) < vtol
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.apply(x)) < vtol
    # FGMRES on J d = F
    F = o.residual()
    h.residual()
    h.copy_residual_to("b")
    its_h, reason_h, _ = h.fgmres("b", "d")
    d_o, its_o, reason_o, _ = la.fgmres(lambda v: la.spmv_block(J, v), o.pc.apply, F, rtol=o.opts["ksp_rtol"],
                                        maxit=o.opts["ksp_max_it"], restart=o.opts["ksp_restart"])
    assert reason_h == reason_o == 2
    assert abs(its_h - its_o) <= 1, (its_h, its_o)
    assert rel2(h.vec_get("d"), d_o) < 1e-6
    h.close()


NEWTON = [
    ("c1", cases.c1_homogeneous, dict(N=12, nphase=1), dict(pc="cpr", ilu_tile=(1 << 30, 64, 1)), 86400.0),
    ("c3", cases.c3_spe10_2d, dict(Nx=14, Ny=19, nphase=2), dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, ilu_tile=(1 << 30, 64, 1)), 864.0),
    ("c4", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=2), dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25), 86.4),
    ("c4_1ph_fscd", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=1), dict(pc="fieldsplit_cd", ksp_rtol=1e-8), 864.0),
    ("c4_cptramg_QI", cases.c4_spe10_3d, dict(Nx=7, Ny=13, Nz=9, nphase=2), dict(pc="cptramg", decoup="QI", ksp_rtol=1e-8, snes_max_it=25), 86.4),
]


@pytest.mark.parametrize("name,builder,kw,opts,dt", NEWTON, ids=[c[0] for c in NEWTON])
def test_newton_parity(name, builder, kw, opts, dt):
    spec, u0, o, h = make(builder, opts, **kw)
    for e in (o, h):
        e.set_state(u0)
    for step in range(2):
        for e in (o, h):
            e.set_old(e.get_state() if e is o else None)
            e.set_dt(dt)
        ro = o.newton_solve()
        rh = h.newton_solve()
        assert ro["reason"] > 0 and rh["reason"] == ro["reason"], (ro, rh)
        assert rh["nits"] == ro["nits"]
        assert abs(rh["lits"] - ro["lits"]) <= max(2, 0.1*ro["lits"])
        uo, uh = o.get_state(), h.get_state()
        assert rel2(uh[0], uo[0]) < 1e-8 and rel2(uh[1], uo[1]) < 1e-8
        if o.b == 3:
            assert np.abs(uh[2] - uo[2]).max() < 1e-8
    h.close()


@pytest.mark.parametrize("mfma", ["1", "0"], ids=["matrix_cores", "scalar"])
def test_coarse_inverse_kernels(mfma):
    """TP_AMG_DENSE_MFMA (read at every set-up): the coarsest-grid inverse by a blocked Gauss-Jordan whose trailing update runs on
    v_mfma_f64_16x16x4_f64 (default) or by the scalar LDS kernel -- the V-cycle and the whole preconditioner against the oracle,
    full (64-cell) and padded coarsest grids."""
    import os
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    os.environ["TP_AMG_DENSE_MFMA"] = mfma
    try:
        for kw in (dict(Nx=16, Ny=18, Nz=16, nphase=2), dict(Nx=9, Ny=14, Nz=8, nphase=2), dict(Nx=7, Ny=5, Nz=3, nphase=1)):
            opts = dict(pc="cptr" if kw["nphase"] == 2 else "cpr")
            spec, u0, *_ = cases.c4_spe10_3d(**kw)
            o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
            u = cases.perturbed_state(spec, seed=5, amp=0.3)
            for e in (o, h):
                e.set_old(u0)
                e.set_dt(8640.0)
                e.set_state(u)
            schur = opts["pc"] == "cptr"
            out = o.jacobian(want_schur=schur)
            J, Sm = out if schur else (out, None)
            h.jacobian()
            o.pc.setup(J, Sm)
            h.pc_setup()
            x = np.random.default_rng(11).standard_normal(u.shape)
            h.vec_set("x", x)
            h.amg_vcycle(0, "x", 0, "y", 0)
            assert rel2(h.vec_get("y")[0], o.pc.amg_p.vcycle(x[0])) < 1e-10, kw
            h.pc_apply("x", "y")
            assert rel2(h.vec_get("y"), o.pc.apply(x)) < 1e-10, kw
            h.close()
    finally:
        del os.environ["TP_AMG_DENSE_MFMA"]


def test_newton_random_boxes_and_presets():
    """Seeded fuzz of whole Newton solves (two time steps each): random boxes, presets and time steps; the same convergence
    reason and Newton count, Krylov counts within 10 %, states to 1e-8 -- through the pipelined FGMRES loop, the speculative
    preconditioner applications included."""
    rng = np.random.default_rng(77)
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    kinds = [dict(pc="cptr"), dict(pc="cpr"), dict(pc="cpr", decoup="QI"), dict(pc="cptr", decoup="QI"), dict(pc="cpr", ilu_levels=1),
             dict(pc="cptramg", decoup="QI"), dict(pc="bilu", ilu_levels=1), dict(pc="cptr", ilu_whole=True, ilu_tile=(4, 3, 3))]
    for it in range(10):
        Nx, Ny, Nz = int(rng.integers(2, 9)), int(rng.integers(3, 15)), int(rng.integers(1, 10))
        opts = dict(kinds[int(rng.integers(0, len(kinds)))], ksp_rtol=1e-8, snes_max_it=25)
        nphase = 2 if opts["pc"] in ("cptr", "cptramg") or rng.integers(0, 2) else 1
        dt = float(rng.choice([43.2, 86.4, 432.0]))
        tag = (it, Nx, Ny, Nz, nphase, opts, dt)
        spec, u0, *_ = cases.c4_spe10_3d(Nx=Nx, Ny=Ny, Nz=Nz, nphase=nphase)
        o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
        for e in (o, h):
            e.set_state(u0)
        for step in range(2):
            for e in (o, h):
                e.set_old(e.get_state() if e is o else None)
                e.set_dt(dt)
            ro, rh = o.newton_solve(), h.newton_solve()
            assert rh["reason"] == ro["reason"], (tag, ro, rh)
            if ro["reason"] <= 0:
                break
            assert rh["nits"] == ro["nits"], (tag, ro, rh)
            assert abs(rh["lits"] - ro["lits"]) <= max(2, 0.1*ro["lits"]), (tag, ro, rh)
            uo, uh = o.get_state(), h.get_state()
            assert rel2(uh[0], uo[0]) < 1e-8 and rel2(uh[1], uo[1]) < 1e-8, tag
            if o.b == 3:
                assert np.abs(uh[2] - uo[2]).max() < 1e-8, tag
        h.close()


def test_ilu_sweeps_random_boxes_and_tiles():
    """Seeded fuzz of the second stage alone (pc_bilu: pc_apply IS the sweep): random box shapes, tile shapes, fill level,
    whole-slab coupling and phase count -- partial tiles, tiles wider than the box, one-cell directions -- against the oracle's
    tiled ILU.  One process, 36 small cases."""
    rng = np.random.default_rng(20261004)
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    for it in range(36):
        Nx, Ny, Nz = (int(v) for v in rng.integers(1, 13, size=3))
        if Nx*Ny*Nz < 2:
            Nz = 3
        nphase = int(rng.integers(1, 3))
        levels = int(rng.integers(0, 2))
        whole = bool(levels == 0 and rng.integers(0, 3) == 0)
        t1, t2 = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        t0 = int(rng.choice([1 << 30, 3, 5, 8]))
        opts = dict(pc="bilu", ilu_levels=levels, ilu_tile=(t0, t1, t2))
        if whole:
            opts["ilu_whole"] = True
        tag = (it, Nx, Ny, Nz, nphase, opts)
        spec, u0, *_ = cases.c4_spe10_3d(Nx=Nx, Ny=Ny, Nz=Nz, nphase=nphase)
        o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
        u = cases.perturbed_state(spec, seed=5 + it, amp=0.3)
        for e in (o, h):
            e.set_old(u0)
            e.set_dt(8640.0)
            e.set_state(u)
        J = o.jacobian()
        h.jacobian()
        o.pc.setup(J)
        h.pc_setup()
        x = rng.standard_normal(u.shape)
        h.vec_set("x", x)
        h.pc_apply("x", "y")
        assert rel2(h.vec_get("y"), o.pc.ilu.solve(x)) < 1e-10, tag
        h.close()


def test_pc_apply_random_boxes_and_cycle_shapes():
    """Seeded fuzz of the whole preconditioner: random box shapes (up to ~8000 cells: several AMG levels, tail, pairs of
    transfer levels), presets, decouplings and cycle-shape options against the oracle."""
    rng = np.random.default_rng(4102026)
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    kinds = [dict(pc="cptr"), dict(pc="cptr", decoup="QI"), dict(pc="cpr", decoup="QI"), dict(pc="cpr"), dict(pc="cptramg", decoup="QI"),
             dict(pc="cptr", schur_a11=True), dict(pc="cpr", ilu_levels=1)]
    for it in range(20):
        Nx, Ny, Nz = int(rng.integers(2, 17)), int(rng.integers(2, 25)), int(rng.integers(1, 21))
        opts = dict(kinds[int(rng.integers(0, len(kinds)))])
        nphase = 2 if opts["pc"] in ("cptramg",) or rng.integers(0, 3) else 1
        if nphase == 1 and opts["pc"] == "cptr":
            opts = dict(pc="cpr", decoup="TI")
        opts.update(amg_full_levels=int(rng.integers(0, 4)), amg_coarse_pre=int(rng.integers(0, 2)), amg_coarse_post=int(rng.integers(1, 3)),
                    amg_mid_skip=bool(rng.integers(0, 2)), amg_tail_post=int(rng.integers(1, 3)), amg_nu=int(rng.integers(1, 3)))
        tag = (it, Nx, Ny, Nz, nphase, opts)
        spec, u0, *_ = cases.c4_spe10_3d(Nx=Nx, Ny=Ny, Nz=Nz, nphase=nphase)
        o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
        u = cases.perturbed_state(spec, seed=50 + it, amp=0.3)
        for e in (o, h):
            e.set_old(u0)
            e.set_dt(8640.0)
            e.set_state(u)
        schur = opts["pc"] == "cptr"
        out = o.jacobian(want_schur=schur)
        J, Sm = out if schur else (out, None)
        h.jacobian()
        o.pc.setup(J, Sm)
        h.pc_setup()
        x = rng.standard_normal(u.shape)
        h.vec_set("x", x)
        h.pc_apply("x", "y")
        assert rel2(h.vec_get("y"), o.pc.apply(x)) < 1e-8, tag
        h.close()


@pytest.mark.parametrize("levels,nphase", [(0, 2), (1, 2), (1, 1)])
def test_bilu_preset_stage(levels, nphase):
    """pc_bilu (twophase.py:758-762, singlephase.py:402-406): bjacobi + block-ILU(levels) alone -- pc_apply IS the sweep,
    FGMRES iteration counts equal the oracle's."""
    import oracle.linalg as la
    spec, u0, o, h = make(cases.c4_spe10_3d, dict(pc="bilu", ilu_levels=levels, ilu_tile=(5, 4, 7)), Nx=9, Ny=14, Nz=8, nphase=nphase)
    u = cases.perturbed_state(spec, seed=5, amp=0.3)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(864.0)
        e.set_state(u)
    J = o.jacobian()
    h.jacobian()
    o.pc.setup(J)
    h.pc_setup()
    x = np.random.default_rng(11).standard_normal(u.shape)
    h.vec_set("x", x)
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.ilu.solve(x)) < 1e-10
    F = o.residual()
    h.residual()
    h.copy_residual_to("b")
    its_h, reason_h, _ = h.fgmres("b", "d")
    d_o, its_o, reason_o, _ = la.fgmres(lambda v: la.spmv_block(J, v), o.pc.apply, F, rtol=o.opts["ksp_rtol"],
                                        maxit=o.opts["ksp_max_it"], restart=o.opts["ksp_restart"])
    assert reason_h == reason_o == 2 and abs(its_h - its_o) <= 1, (its_h, its_o)
    assert rel2(h.vec_get("d"), d_o) < 1e-6
    h.close()


@pytest.mark.parametrize("env", [{"TP_ILU_MW": "0"}, {"TP_ILU_MW": "0", "TP_ILU_YLDS": "0"}, {"TP_ILU_BLOCK": "0"}, {"TP_ILU_BLOCK": "1"},
                                 {"TP_ILU_YLDS": "0"}, {"TP_ILU_YLDS": "0", "TP_ILU_BLOCK": "1"}, {"TP_ASM_LDS": "1"},
                                 {"TP_ILU1_PACK": "0"}, {"TP_ILU1_FACTOR_TILE": "0", "TP_ILU1_PF": "2"},
                                 {"TP_ILU1_PACK": "0", "TP_ILU1_FACTOR_TILE": "0"},
                                 {"TP_BAMG_TAIL_CELLS": "0", "TP_BAMG_FUSE_BELOW": "0", "TP_BAMG_DENSE_LDS": "0"}],
                         ids=["one_wave", "one_wave_y_hbm", "mw_per_step", "mw_blocks", "mw_y_hbm", "mw_y_hbm_blocks", "asm_lds_tiled",
                              "ilu1_padded_stream", "ilu1_factor_per_step_launches", "ilu1_padded_per_step", "system_amg_per_level_kernels"])
def test_env_selected_sweep_kernels(env):
    """The ILU(0) sweep kernels that are not the default of a given grid -- the one-wave kernel, the multi-wave kernel with /
    without block transfers and with y through HBM -- are selected by environment variables the library reads once per
    process: each runs tests/ilu_env_check.py (sweeps and whole preconditioner vs the oracle, 3-D and 2-D, partial tiles) in
    ONE child process."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "ilu_env_check.py")], env={**os.environ, **env},
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-2000:])


def test_exported_vector_ops():
    """tp_vec_dot_batch / tp_vec_axpy_batch / tp_vec_norm2 (VecMDot, VecMAXPY, VecNorm of one Krylov iteration) vs numpy."""
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=9, Ny=14, Nz=8, nphase=2)
    h = HipEngine(spec, dict(pc="cptr"))
    rng = np.random.default_rng(4)
    n = 5
    V = rng.standard_normal((n, 3) + spec["phi"].shape)
    w = rng.standard_normal((3,) + spec["phi"].shape)
    h.vec_batch("v", n)
    for i in range(n):
        h.vec_set("v%d" % i, V[i])
    h.vec_set("w", w)
    d = h.dot_batch("v", n, "w")
    assert np.allclose(d, [np.vdot(V[i], w) for i in range(n)], rtol=1e-12, atol=1e-9)
    assert abs(h.norm2("w") - np.linalg.norm(w)) <= 1e-12*np.linalg.norm(w)
    coef = rng.standard_normal(n)
    h.axpy_batch("v", n, coef, "w")
    assert rel2(h.vec_get("w"), w + np.tensordot(coef, V, axes=1)) < 1e-13
    with pytest.raises(Exception):
        h.vec("solo")
        h._ck(h.lib.tp_vec_dot_batch(h.ctx, h.vec("solo"), 2, h.vec("w"), None))
    h.close()


def test_ksp_residual_monitor_per_field():
    """tp_set_ksp_monitor (the reference's ksp_monitor_residuals, thermalmodel.py:44-74): per-field norms of the TRUE
    residual b - J x_j at every FGMRES iteration; they follow the recurrence norm and end at ||b - J d||."""
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=9, Ny=14, Nz=8, nphase=2)
    h = HipEngine(spec, dict(pc="cptr", ksp_rtol=1e-8))
    u = cases.perturbed_state(spec, seed=5, amp=0.3)
    h.set_old(u0)
    h.set_dt(8640.0)
    h.set_state(u)
    h.jacobian()
    h.residual()
    h.copy_residual_to("b")
    seen = []
    h.set_ksp_monitor(lambda its, rn, fn: seen.append((its, rn, list(fn))))
    its, reason, rnorm = h.fgmres("b", "d")
    h.set_ksp_monitor(None)
    assert reason == 2 and len(seen) == its and [s[0] for s in seen] == list(range(1, its + 1))
    for it, rn, fn in seen:
        assert len(fn) == 3 and abs(np.sqrt(sum(v*v for v in fn)) - rn) <= 1e-6*seen[0][1] + 1e-3*rn
    h.spmv("d", "Jd")
    r = h.vec_get("b") - h.vec_get("Jd")
    for f in range(3):
        assert abs(np.linalg.norm(r[f]) - seen[-1][2][f]) <= 1e-9*np.linalg.norm(h.vec_get("b"))
    n0 = len(seen)
    h.fgmres("b", "d")                      # monitor removed: no more calls
    assert len(seen) == n0
    h.close()


@pytest.mark.parametrize("grid,dt,expect", [((7, 13, 9), 86.4, 0), ((20, 26, 18), 86.4, 0), ((20, 26, 18), 4.0e6, -1),
                                            ((20, 26, 18), 86.4, None)],
                         ids=["tail_truncated", "big_truncated", "not_dominant", "tau_off"])
def test_dominance_truncated_hierarchy(grid, dt, expect):
    """amg_dom_tau: at small dt the temperature operator S~ is strongly diagonally dominant and its hierarchy ends on
    level 0 with two Jacobi sweeps (inside the single-workgroup tail on the small grid, as a streaming kernel on the
    larger one); at a huge dt it is not and the full V-cycle runs; the pressure hierarchy is never truncated.  GPU and
    oracle take the same decision and agree on every stage."""
    opts = dict(pc="cptr", ksp_rtol=1e-8)
    if expect is None:
        opts["amg_dom_tau"] = 0.0
    spec, u0, o, h = make(cases.c4_spe10_3d, opts, Nx=grid[0], Ny=grid[1], Nz=grid[2], nphase=2)
    u = cases.perturbed_state(spec, seed=5, amp=0.05)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(dt)
        e.set_state(u)
    J, Sm = o.jacobian(want_schur=True)
    h.jacobian()
    o.pc.setup(J, Sm)
    h.pc_setup()
    tT, r0 = h.amg_trunc(1)
    want = -1 if expect is None else expect
    assert tT == want == (-1 if o.pc.amg_T.trunc is None else o.pc.amg_T.trunc), (tT, r0, o.pc.amg_T.trunc)
    assert h.amg_trunc(0)[0] == -1 and o.pc.amg_p.trunc is None
    if expect is not None:
        ratio = float((np.abs(Sm[1:]).sum(axis=0)/np.abs(Sm[0])).max())
        assert abs(r0 - ratio) <= 1e-12*ratio
    x = np.random.default_rng(11).standard_normal(u.shape)
    h.vec_set("x", x)
    h.amg_vcycle(1, "x", 1, "y", 1)
    assert rel2(h.vec_get("y")[1], o.pc.amg_T.vcycle(x[1])) < 1e-10
    h.pc_apply("x", "y")
    # TOLERANCE: 1e-10 is the bar of every PC application in this file; the one dt = 46-day case gets 1e-9 because its
    # stage-1 operators have condition number ~1e6 (accumulation term ~1/dt gone, pure elliptic pressure): the measured
    # GPU-vs-oracle difference there is 1.9e-10 = eps * cond * O(1), i.e. summation order, not an algorithmic difference
    assert rel2(h.vec_get("y"), o.pc.apply(x)) < (1e-9 if dt > 1e5 else 1e-10)
    if dt < 1e5:                  # (at dt = 46 days from a 5 % perturbed state FGMRES(200) stalls in every engine)
        import oracle.linalg as la
        F = o.residual()
        h.residual()
        h.copy_residual_to("b")
        its_h, reason_h, _ = h.fgmres("b", "d")
        d_o, its_o, reason_o, _ = la.fgmres(lambda v: la.spmv_block(J, v), o.pc.apply, F, rtol=1e-8)
        assert reason_h == reason_o == 2 and abs(its_h - its_o) <= 1
    h.close()
