"""GPU tests at the model level (SinglePhase/TwoPhase/PC classes through the C ABI) and size-independent
properties at BASELINE config 4's full size."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def rel2(a, b):
    return np.linalg.norm((a - b).ravel())/max(np.linalg.norm(b.ravel()), 1e-300)


def test_time_loop_parity_config1():
    """BASELINE config 1 end to end (tests/test_homo_wells.py of the reference: 2 steps, const-rate wells,
    pc_cpr) through the SinglePhase/ThermalModel API: HIP engine vs oracle engine."""
    from oracle.engine import OracleEngine
    from thermalporous_amd.singlephase import SinglePhase
    res = []
    for factory in (OracleEngine, None):
        spec, u0, p, g, c = cases.c1_homogeneous(N=20)
        m = SinglePhase(g, c, p, end=2.0, maxdt=1.0, small_dt_start=False, solver_parameters="pc_cpr",
                        filename=None, verbosity=False, _engine_factory=factory)
        m.solve()
        res.append((m.nits_vec, m.lits_vec, m.u.dat.data_ro[0].copy(), m.u.dat.data_ro[1].copy()))
    assert res[0][0] == res[1][0]
    assert all(abs(a - b) <= 1 for a, b in zip(res[0][1], res[1][1]))
    assert rel2(res[1][2], res[0][2]) < 1e-8 and rel2(res[1][3], res[0][3]) < 1e-8


def test_time_loop_parity_two_phase_2d():
    """BASELINE config 3 (reduced): two-phase 2-D SPE10-like layer, Peaceman wells, pc_cptr, adaptive dt."""
    from oracle.engine import OracleEngine
    from thermalporous_amd.twophase import TwoPhase
    res = []
    for factory in (OracleEngine, None):
        spec, u0, p, g, c = cases.c3_spe10_2d(16, 22, 2)
        m = TwoPhase(g, c, p, end=0.004, maxdt=0.002, solver_parameters="pc_cptr", filename=None, verbosity=False,
                     _engine_factory=factory)
        m.solve()
        res.append((m.nits_vec, m.lits_vec, m.dt_vec, [m.u.dat.data_ro[f].copy() for f in range(3)]))
    assert res[0][0] == res[1][0] and np.allclose(res[0][2], res[1][2])
    assert sum(abs(a - b) for a, b in zip(res[0][1], res[1][1])) <= max(3, 0.05*sum(res[0][1]))
    for f in range(2):
        assert rel2(res[1][3][f], res[0][3][f]) < 1e-7
    assert np.abs(res[1][3][2] - res[0][3][2]).max() < 1e-7


def test_python_pc_classes_reproduce_native_pc_apply():
    """Driving CPTRStage1PC + ILU through the PCBase-shaped Python classes == tp_pc_apply."""
    from thermalporous_amd import preconditioners as pcs
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=7, Ny=9, Nz=6)
    h = HipEngine(spec, dict(pc="cptr"))
    u = cases.perturbed_state(spec, seed=9, amp=0.3)
    h.set_old(u0)
    h.set_dt(4000.0)
    h.set_state(u)
    h.jacobian()
    pc = pcs.PC(h, {"decoup": "No", "vector": False}, prefix="sub_0_")
    comp = pcs.CompositePC(pcs.CPTRStage1PC())
    comp.setUp(pc)
    x = np.random.default_rng(2).standard_normal((3,) + spec["phi"].shape)
    h.vec_set("x", x)
    comp.apply(pc, "x", "y_py")
    h.pc_apply("x", "y_native")
    assert rel2(h.vec_get("y_py"), h.vec_get("y_native")) < 1e-12
    # ConvDiffSchurTwoPhasesPC.apply = one V-cycle on S~ (temperature field)
    schur = pcs.ConvDiffSchurTwoPhasesPC()
    schur.setUp(pc)
    schur.apply(pc, "x", "ys")
    h.amg_vcycle(1, "x", 1, "yv", 1)
    assert rel2(h.vec_get("ys")[1], h.vec_get("yv")[1]) < 1e-14
    h.close()


def test_fieldsplit_cd_preset_and_pc_classes():
    """The single-phase pc_fieldsplit_cd preset (singlephase.py:309-319) through SinglePhase.solve(): HIP engine vs
    oracle engine; and the PCBase-shaped ConvDiffSchurPC / FieldsplitSchurPC objects reproduce tp_pc_apply."""
    from oracle.engine import OracleEngine
    from thermalporous_amd import preconditioners as pcs
    from thermalporous_amd.engine import HipEngine
    from thermalporous_amd.singlephase import SinglePhase
    res = []
    for factory in (OracleEngine, None):
        spec, u0, p, g, c = cases.c1_homogeneous(N=16)
        m = SinglePhase(g, c, p, end=2.0, maxdt=1.0, small_dt_start=False, solver_parameters="pc_fieldsplit_cd",
                        filename=None, verbosity=False, _engine_factory=factory)
        assert m.engine_opts["pc"] == "fieldsplit_cd"
        m.solve()
        res.append((m.nits_vec, m.lits_vec, m.u.dat.data_ro[0].copy(), m.u.dat.data_ro[1].copy()))
    assert res[0][0] == res[1][0]
    assert all(abs(a - b) <= 1 for a, b in zip(res[0][1], res[1][1]))
    assert rel2(res[1][2], res[0][2]) < 1e-8 and rel2(res[1][3], res[0][3]) < 1e-8

    spec, u0, *_ = cases.c4_spe10_3d(Nx=7, Ny=9, Nz=6, nphase=1)
    h = HipEngine(spec, dict(pc="fieldsplit_cd"))
    u = cases.perturbed_state(spec, seed=9, amp=0.3)
    h.set_old(u0)
    h.set_dt(4000.0)
    h.set_state(u)
    h.jacobian()
    pc = pcs.PC(h, {"decoup": "No"}, prefix="fieldsplit_1_")
    fs = pcs.FieldsplitSchurPC(pcs.ConvDiffSchurPC())
    fs.setUp(pc)
    x = np.random.default_rng(2).standard_normal((2,) + spec["phi"].shape)
    h.vec_set("x", x)
    fs.apply(pc, "x", "y_py")
    h.pc_apply("x", "y_native")
    assert rel2(h.vec_get("y_py"), h.vec_get("y_native")) < 1e-12
    with pytest.raises(NotImplementedError):
        pcs.ConvDiffSchurTwoPhasesPC().setUp(pc)      # the two-phase S~ belongs to pc_cptr
    h.close()


@pytest.mark.parametrize("preset", ["pc_fieldsplit_selfp", "pc_fieldsplit_a11", "pc_fieldsplit_diag", "pc_bilu"])
def test_single_phase_schur_presets_time_loop(preset):
    """pc_fieldsplit_selfp (singlephase.py:322-330), pc_fieldsplit_a11 (:331-338), pc_fieldsplit_diag (:371-375) and pc_bilu
    (:402-406) through SinglePhase.solve() on the
    SPE10-like 2-D case: HIP engine vs oracle engine, same Newton counts, Krylov counts +-1, same converged state."""
    from oracle.engine import OracleEngine
    from thermalporous_amd.singlephase import SinglePhase
    res = []
    for factory in (OracleEngine, None):
        spec, u0, p, g, c = cases.c3_spe10_2d(Nx=14, Ny=19, nphase=1)
        # (the block-diagonal preset is a weak preconditioner: at dt = 0.25 d its fourth Newton iterate asks FGMRES for a
        # residual below what float64 can deliver on this matrix -- in the oracle as on the GPU -- so it runs at dt = 0.01 d)
        end, maxdt = (0.02, 0.01) if preset == "pc_fieldsplit_diag" else (0.5, 0.25)
        m = SinglePhase(g, c, p, end=end, maxdt=maxdt, small_dt_start=False, solver_parameters=preset, filename=None,
                        verbosity=False, _engine_factory=factory)
        assert m.engine_opts["schur_selfp"] == (preset == "pc_fieldsplit_selfp")
        assert m.engine_opts["fs_additive"] == (preset == "pc_fieldsplit_diag")
        m.solve()
        res.append((m.nits_vec, m.lits_vec, m.u.dat.data_ro[0].copy(), m.u.dat.data_ro[1].copy()))
    assert res[0][0] == res[1][0] and len(res[0][0]) >= 2, (res[0][0], res[1][0])
    # TOLERANCE: Krylov counts +-1 -- except pc_fieldsplit_diag, +-2: the block-diagonal preconditioner leaves FGMRES creeping
    # towards rtol over its last iterations (residual reduction < 5 % per iteration), so whether the threshold is crossed at
    # iteration 17 or 19 is decided by round-off (measured with the round-3 default amg_omega = 0.9: oracle 19, GPU 17 in one of
    # the three solves; Newton counts and the converged states still agree to the bars below)
    slack = 2 if preset == "pc_fieldsplit_diag" else 1
    assert all(abs(a - b) <= slack for a, b in zip(res[0][1], res[1][1])), (res[0][1], res[1][1])
    assert rel2(res[1][2], res[0][2]) < 1e-8 and rel2(res[1][3], res[0][3]) < 1e-8


def test_two_phase_ilu1_preset_time_loop():
    """pc_cprilu1_gmres (twophase.py:653-668: CPR stage 1 + block-ILU(1) second stage) through TwoPhase.solve():
    HIP engine vs oracle engine."""
    from oracle.engine import OracleEngine
    from thermalporous_amd.twophase import TwoPhase
    res = []
    for factory in (OracleEngine, None):
        spec, u0, p, g, c = cases.c4_spe10_3d(Nx=7, Ny=13, Nz=9, nphase=2)
        m = TwoPhase(g, c, p, end=0.02, maxdt=0.01, small_dt_start=False, solver_parameters="pc_cprilu1_gmres",
                     filename=None, verbosity=False, _engine_factory=factory)
        assert m.engine_opts["ilu_levels"] == 1 and m.engine_opts["pc"] == "cpr"
        m.solve()
        res.append((m.nits_vec, m.lits_vec, [d.copy() for d in m.u.dat.data_ro]))
    assert res[0][0] == res[1][0] and len(res[0][0]) >= 2
    assert all(abs(a - b) <= 1 for a, b in zip(res[0][1], res[1][1]))
    for a, b in zip(res[1][2], res[0][2]):
        assert rel2(a, b) < 1e-7


def test_two_phase_cptr_a11_preset_time_loop():
    """pc_cptr_a11 (twophase.py:598-616) AS WRITTEN in the reference -- Schur complement preconditioned by A_11 and
    ``sub_1_pc_bjacobi_blocks: 1`` (:612): one bjacobi block = whole-grid ILU(0) -- through TwoPhase.solve() on the HIP engine
    (whole-slab ILU(0) as tile-diagonal sweeps, tp_options.ilu_whole) vs the oracle engine (one tile = the grid).  The grid
    has 13 x 9 = 117 columns: more than one wavefront, so the round-2 engine raised NotImplementedError here."""
    from oracle.engine import OracleEngine
    from thermalporous_amd.twophase import TwoPhase
    res = []
    for factory in (OracleEngine, None):
        spec, u0, p, g, c = cases.c4_spe10_3d(Nx=13, Ny=17, Nz=9, nphase=2)
        m = TwoPhase(g, c, p, end=0.02, maxdt=0.01, small_dt_start=False, solver_parameters="pc_cptr_a11",
                     filename=None, verbosity=False, _engine_factory=factory)
        assert m.engine_opts["bjacobi_blocks"] == 1 and m.engine_opts["schur_a11"] is True and m.engine_opts["pc"] == "cptr"
        m.solve()
        if factory is None:
            assert m.engine.opts["ilu_whole"] is True
        res.append((m.nits_vec, m.lits_vec, [d.copy() for d in m.u.dat.data_ro]))
    assert res[0][0] == res[1][0] and len(res[0][0]) >= 2
    assert all(abs(a - b) <= 1 for a, b in zip(res[0][1], res[1][1]))
    for a, b in zip(res[1][2], res[0][2]):
        assert rel2(a, b) < 1e-7


def test_full_size_properties_c4():
    """BASELINE config 4 at full size (60x220x85): size-independent properties of the GPU path --
    pairwise cancellation of the face fluxes, linearity of every preconditioner stage, and the true
    residual of the FGMRES solution."""
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(60, 220, 85)
    nosrc = dict(spec)
    nosrc["sources"] = None
    h = HipEngine(nosrc, dict(pc="cptr", ksp_rtol=1e-8))
    u = cases.perturbed_state(spec, seed=1, amp=0.05)
    h.set_old(u)
    h.set_state(u)
    # u == u_old: the accumulation vanishes, the residual is the sum of face fluxes -> each field sums to 0
    h.set_dt(50.0)
    Rflux = h.residual()
    for f in range(3):
        assert abs(Rflux[f].sum()) <= 1e-9*np.abs(Rflux[f]).sum()
    h.set_old(u0)
    h.set_state(u)
    h.jacobian()
    h.pc_setup()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(u.shape)
    y = rng.standard_normal(u.shape)
    for apply in (h.ilu_solve, h.stage1_apply, h.pc_apply):     # M(2x - 3y) = 2 Mx - 3 My
        h.vec_set("x", x)
        apply("x", "mx")
        mx = h.vec_get("mx")
        h.vec_set("x", y)
        apply("x", "my")
        my = h.vec_get("my")
        h.vec_set("x", 2.0*x - 3.0*y)
        apply("x", "mz")
        assert rel2(h.vec_get("mz"), 2.0*mx - 3.0*my) < 1e-9
    h.residual()
    h.copy_residual_to("b")
    its, reason, rn = h.fgmres("b", "d")
    assert reason == 2 and its < 100
    h.spmv("d", "Jd")
    b = h.vec_get("b")
    assert np.linalg.norm(h.vec_get("Jd") - b) <= 1.5e-8*np.linalg.norm(b)
    h.close()


def test_time_kernel_before_pc_setup_single_phase_fieldsplit():
    """Regression (round-1 advisor): on a 2-field context tp_time_kernel used to allocate the scratch vector w3 with
    b*ntot = 2 planes while the Schur stage writes a third one.  Timing a kernel BEFORE any pc_setup and then applying
    the preconditioner must give the same result as a context that never called tp_time_kernel."""
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=9, Ny=11, Nz=7, nphase=1)
    u = cases.perturbed_state(spec, seed=4, amp=0.3)
    x = np.random.default_rng(5).standard_normal(u.shape)
    res = []
    for timed_first in (True, False):
        h = HipEngine(spec, dict(pc="fieldsplit_cd"))
        h.set_old(u0)
        h.set_dt(4000.0)
        h.set_state(u)
        h.jacobian()
        if timed_first:
            for which in (0, 3, 5, 6):
                assert h.time_kernel(which, 2) > 0.0
            h.jacobian()
        h.pc_setup()
        h.vec_set("x", x)
        h.pc_apply("x", "y")
        res.append(h.vec_get("y"))
        h.close()
    assert rel2(res[0], res[1]) < 1e-13


def test_time_kernel_pc_apply_after_a_solve():
    """tp_time_kernel(which=4) right after Newton solves: the first capture of a NEW (input, output) hipGraph pair
    after cached pairs have been replayed (the call that crashed once under rocprofv3) -- and the
    captured application equals the eager one."""
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=12, Ny=20, Nz=10, nphase=2)
    h = HipEngine(spec, dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25))
    h.set_state(u0)
    for dt in (8.0, 16.0):
        h.set_old(None)
        h.set_dt(dt)
        assert h.newton_solve()["reason"] > 0
    h.jacobian()
    h.pc_setup()
    for which in (0, 1, 2, 3, 4, 5, 6, 4):
        assert h.time_kernel(which, 3) > 0.0
    x = np.random.default_rng(6).standard_normal((3,) + spec["phi"].shape)
    h.vec_set("x", x)
    h.pc_apply("x", "y1")                 # captured
    h.stage1_apply("x", "s")              # eager pieces: y = s + ILU(x - J s)
    h.spmv("s", "Js")
    h.vec_axpby("r", 1.0, "x", -1.0, "Js")
    h.ilu_solve("r", "z")
    h.vec_axpby("y2", 1.0, "s", 1.0, "z")
    assert rel2(h.vec_get("y1"), h.vec_get("y2")) < 1e-12
    h.close()
