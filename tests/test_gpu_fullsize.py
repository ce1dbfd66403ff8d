"""GPU parity at the BASELINE configurations' TRUE sizes.

  * C2 / C3 (60x220 2-D, single-phase pc_cpr / two-phase pc_cptr): the numpy oracle at full size.
  * C4 (60x220x85, two-phase pc_cptr): oracle/cport -- the C++/OpenMP twin of the numpy oracle (tests/test_cport.py
    pins it to the numpy oracle to round-off) -- at full size: assembly entries, every linear stage, FGMRES, Newton.
  * C5 (240x880x340 on 8 GPUs): ONE of its eight slabs (240x110x340, 9M cells, what each GPU owns) through the
    size-independent properties, and the 2-slab algorithm on a box that fits one GPU twice.
Tolerances as in tests/test_gpu_parity.py.  Parity with the reference itself is unpinned (oracle/__init__.py)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def rel2(a, b):
    return np.linalg.norm((a - b).ravel())/max(np.linalg.norm(b.ravel()), 1e-300)


def _check_assembly(o, h, schur):
    Ro, Rh = o.residual(), h.residual()
    for f in range(h.b):
        assert np.abs(Rh[f] - Ro[f]).max() <= 1e-11*np.abs(Ro[f]).max(), ("residual field", f)
    out_o, out_h = o.jacobian(want_schur=schur), h.jacobian(want_schur=schur)
    Jo, Jh = (out_o[0], out_h[0]) if schur else (out_o, out_h)
    for r in range(h.b):
        for c in range(h.b):
            scale = np.abs(Jo[:, r, c]).max()
            if scale == 0.0:
                assert np.abs(Jh[:, r, c]).max() == 0.0
                continue
            assert np.abs(Jh[:, r, c] - Jo[:, r, c]).max()/scale < 1e-11, ("J block", r, c)
    if schur:
        assert np.abs(out_h[1] - out_o[1]).max() <= 1e-11*np.abs(out_o[1]).max()
    return (Jo, out_o[1]) if schur else (Jo, None)


@pytest.mark.parametrize("nphase,opts,dts", [(1, dict(pc="cpr"), (84.375, 168.75, 337.5)),
                                             (2, dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25), (21.09375, 42.1875, 84.375))],
                         ids=["c2_1ph_cpr", "c3_2ph_cptr"])
def test_c2_c3_true_size_60x220(nphase, opts, dts):
    """BASELINE configs 2 and 3 at 60x220 with the engine's default tiles: assembly, stages, FGMRES, two Newton
    solves of the dt ramp's first steps -- HIP vs the numpy oracle at the same size."""
    import oracle.linalg as la
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c3_spe10_2d(Nx=60, Ny=220, nphase=nphase)
    o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
    schur = opts["pc"] == "cptr"
    u = cases.perturbed_state(spec, seed=3, amp=0.3)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(8640.0)
        e.set_state(u)
    J, Sm = _check_assembly(o, h, schur)
    o.pc.setup(J, Sm)
    h.pc_setup()
    x = np.random.default_rng(11).standard_normal(J.shape[1:2] + J.shape[3:])
    h.vec_set("x", x)
    h.spmv("x", "y")
    assert rel2(h.vec_get("y"), la.spmv_block(J, x)) < 1e-12
    h.ilu_solve("x", "y")
    assert rel2(h.vec_get("y"), o.pc.ilu.solve(x)) < 1e-10
    h.stage1_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.stage1(x)) < 1e-9
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.apply(x)) < 1e-9
    # three doubling time steps from the uniform state (C2: the ramp's own first steps, maxdt 1 day * 2^-10; C3: two
    # octaves lower -- with rate 2e-4 the step 84 s -> 169 s does not converge in either engine and the time loop chops)
    for e in (o, h):
        e.set_state(u0)
    for dt in dts:
        for e in (o, h):
            e.set_old(e.get_state() if e is o else None)
            e.set_dt(dt)
        ro, rh = o.newton_solve(), h.newton_solve()
        # the tiny first steps are almost linear: the second Newton iterate sits AT the convergence thresholds
        # (||F|| <= 1e-8 ||F0|| vs ||dx|| < 1e-8 ||x||), so which test fires first, and with it the last iteration,
        # may differ by rounding; both must converge to the same state within the solver tolerance
        # TOLERANCES here (Newton +-1, states 1e-7 instead of the file-wide "equal" / 1e-8): a run that stops one Newton
        # iteration earlier holds a state whose error is the LAST update, bounded by the stol/rtol tests at 1e-8 relative
        # to ||x|| per field but up to ~1e-7 in the 2-norm of a single field; cases that take the same number of
        # iterations agree to 1e-9 (test_c4_true_size_vs_cport checks that bar)
        assert ro["reason"] > 0 and rh["reason"] > 0, (ro, rh)
        assert abs(rh["nits"] - ro["nits"]) <= 1
        assert abs(rh["lits"] - ro["lits"]) <= max(2, 0.15*ro["lits"])
        uo, uh = o.get_state(), h.get_state()
        for f in range(2):
            assert rel2(uh[f], uo[f]) < 1e-7
        if h.b == 3:
            assert np.abs(uh[2] - uo[2]).max() < 1e-7
    h.close()


def test_c4_true_size_vs_cport():
    """BASELINE config 4 at 60x220x85 (1 122 000 cells): every stage of the HIP path against oracle/cport."""
    from oracle.cport import CPortEngine
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(60, 220, 85)
    opts = dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25)
    c, h = CPortEngine(spec, opts), HipEngine(spec, opts)
    u = cases.perturbed_state(spec, seed=1, amp=0.05)
    for e in (c, h):
        e.set_old(u0)
        e.set_dt(600.0)
        e.set_state(u)
    _check_assembly(c, h, True)
    c.pc_setup()
    h.pc_setup()
    x = np.random.default_rng(11).standard_normal(u.shape)
    h.vec_set("x", x)
    h.spmv("x", "y")
    assert rel2(h.vec_get("y"), c.spmv(x)) < 1e-12
    h.ilu_solve("x", "y")
    assert rel2(h.vec_get("y"), c.ilu_solve(x)) < 1e-10
    h.amg_vcycle(0, "x", 0, "y", 0)
    assert rel2(h.vec_get("y")[0], c.vcycle(0, x[0])) < 1e-9
    h.amg_vcycle(1, "x", 1, "y", 1)
    assert rel2(h.vec_get("y")[1], c.vcycle(1, x[1])) < 1e-9
    h.stage1_apply("x", "y")
    assert rel2(h.vec_get("y"), c.stage1(x)) < 1e-9
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), c.pc_apply(x)) < 1e-9
    F = c.residual()
    h.residual()
    h.copy_residual_to("b")
    its_h, reason_h, _ = h.fgmres("b", "d")
    d_c, its_c, reason_c, _ = c.fgmres(F)
    assert reason_h == reason_c == 2 and abs(its_h - its_c) <= 1, (its_h, its_c)
    assert rel2(h.vec_get("d"), d_c) < 1e-6
    # one Newton solve (a time step of the ramp) from the uniform initial state
    for e in (c, h):
        e.set_state(u0)
        e.set_old(None)
        e.set_dt(8.4375)
    rc, rh = c.newton_solve(), h.newton_solve()
    assert rc["reason"] > 0 and rh["reason"] == rc["reason"] and rh["nits"] == rc["nits"], (rc, rh)
    assert abs(rh["lits"] - rc["lits"]) <= max(2, 0.1*rc["lits"])
    uc, uh = c.get_state(), h.get_state()
    assert rel2(uh[0], uc[0]) < 1e-8 and rel2(uh[1], uc[1]) < 1e-8 and np.abs(uh[2] - uc[2]).max() < 1e-8
    h.close()


def test_c2_true_size_selfp_60x220():
    """pc_fieldsplit_selfp (singlephase.py:322-330) at BASELINE config 2's size: the Sp collapse / exact-Sp sweep stage,
    the whole Schur-FULL preconditioner and FGMRES, HIP vs the numpy oracle."""
    import oracle.linalg as la
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c3_spe10_2d(Nx=60, Ny=220, nphase=1)
    opts = dict(pc="fieldsplit_cd", schur_selfp=True, ksp_rtol=1e-8)
    o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
    u = cases.perturbed_state(spec, seed=3, amp=0.3)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(8640.0)
        e.set_state(u)
    J, Sm = _check_assembly(o, h, True)
    o.pc.setup(J, Sm)
    h.pc_setup()
    x = np.random.default_rng(11).standard_normal(J.shape[1:2] + J.shape[3:])
    h.vec_set("x", x)
    h.amg_vcycle(1, "x", 1, "y", 1)                  # V7: the hierarchy of Sp's 7-point collapse
    assert rel2(h.vec_get("y")[1], o.pc.selfp.amg.vcycle(x[1])) < 1e-9
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.apply(x)) < 1e-9
    F = o.residual()
    h.residual()
    h.copy_residual_to("b")
    its_h, reason_h, _ = h.fgmres("b", "d")
    d_o, its_o, reason_o, _ = la.fgmres(lambda v: la.spmv_block(J, v), o.pc.apply, F, rtol=1e-8)
    assert reason_h == reason_o == 2 and abs(its_h - its_o) <= 1, (its_h, its_o)
    assert rel2(h.vec_get("d"), d_o) < 1e-6
    h.close()


def test_c4_true_size_ilu1_vs_cport():
    """Block-ILU(1) second stage (pc_cprilu1_gmres, twophase.py:653-668) at BASELINE config 4's size, default tiles
    (250 tiles of 85 x 6 x 9 cells, 127 wavefront steps): sweeps, whole preconditioner, FGMRES -- HIP vs oracle/cport."""
    from oracle.cport import CPortEngine
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(60, 220, 85)
    opts = dict(pc="cpr", ilu_levels=1, ksp_rtol=1e-8, snes_max_it=25)
    c, h = CPortEngine(spec, opts), HipEngine(spec, opts)
    u = cases.perturbed_state(spec, seed=1, amp=0.05)
    for e in (c, h):
        e.set_old(u0)
        e.set_dt(600.0)
        e.set_state(u)
    c.residual()
    c.jacobian()
    h.jacobian()
    c.pc_setup()
    h.pc_setup()
    x = np.random.default_rng(11).standard_normal(u.shape)
    h.vec_set("x", x)
    h.ilu_solve("x", "y")
    assert rel2(h.vec_get("y"), c.ilu_solve(x)) < 1e-10
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), c.pc_apply(x)) < 1e-9
    F = c.residual()
    h.residual()
    h.copy_residual_to("b")
    its_h, reason_h, _ = h.fgmres("b", "d")
    d_c, its_c, reason_c, _ = c.fgmres(F)
    assert reason_h == reason_c == 2 and abs(its_h - its_c) <= 1, (its_h, its_c)
    assert rel2(h.vec_get("d"), d_c) < 1e-6
    h.close()


def test_whole_slab_ilu0_true_sizes_c2_c4():
    """``sub_1_pc_bjacobi_blocks: 1`` (tests/test_homo_wells.py:112 of the reference; inside pc_cptr_a11, twophase.py:612): ONE
    bjacobi block = block-ILU(0) of the whole grid, on the GPU as tile-diagonal sweeps (tp_options.ilu_whole).  At the TRUE
    sizes of BASELINE configs 2 (60x220, numpy oracle) and 4 (60x220x85, oracle/cport): the sweep and the whole preconditioner
    equal the one-block oracle, FGMRES counts agree, and the whole-grid factorisation needs no more Krylov iterations than the
    default tiles on the same system (the -5 % of DESIGN.md 4.4 ii, now measured on the GPU)."""
    import oracle.linalg as la
    from oracle.cport import CPortEngine
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    # C2: single-phase 60x220, pc_cpr, one block (the reference's own config-1 dictionary on the SPE10 layer)
    spec, u0, *_ = cases.c3_spe10_2d(Nx=60, Ny=220, nphase=1)
    opts = dict(pc="cpr", decoup="QI", ksp_rtol=1e-8, bjacobi_blocks=1)
    o, h = OracleEngine(spec, opts), HipEngine(spec, opts)
    assert tuple(o.opts["ilu_tile"]) == (60, 220, 1) and h.opts["ilu_whole"]
    u = cases.perturbed_state(spec, seed=3, amp=0.3)
    for e in (o, h):
        e.set_old(u0)
        e.set_dt(8640.0)
        e.set_state(u)
    J = o.jacobian()
    h.jacobian()
    o.pc.setup(J)
    h.pc_setup()
    x = np.random.default_rng(11).standard_normal(u.shape)
    h.vec_set("x", x)
    h.ilu_solve("x", "y")
    assert rel2(h.vec_get("y"), o.pc.ilu.solve(x)) < 1e-10
    h.pc_apply("x", "y")
    assert rel2(h.vec_get("y"), o.pc.apply(x)) < 1e-9
    F = o.residual()
    h.residual()
    h.copy_residual_to("b")
    its_h, reason_h, _ = h.fgmres("b", "d")
    d_o, its_o, reason_o, _ = la.fgmres(lambda v: la.spmv_block(J, v), o.pc.apply, F, rtol=1e-8)
    assert reason_h == reason_o == 2 and abs(its_h - its_o) <= 1, (its_h, its_o)
    h.close()
    # C4: two-phase 60x220x85, pc_cptr, one block vs the default 6x9 tiles
    spec, u0, *_ = cases.c4_spe10_3d(60, 220, 85)
    u = cases.perturbed_state(spec, seed=1, amp=0.05)
    x = np.random.default_rng(11).standard_normal(u.shape)
    its = {}
    for name, extra in (("tiles", {}), ("whole", dict(bjacobi_blocks=1))):
        opts = dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, **extra)
        c, h = CPortEngine(spec, opts), HipEngine(spec, opts)
        for e in (c, h):
            e.set_old(u0)
            e.set_dt(600.0)
            e.set_state(u)
        c.residual()
        c.jacobian()
        h.jacobian()
        c.pc_setup()
        h.pc_setup()
        if name == "whole":
            assert c.ntiles() == 1
            h.vec_set("x", x)
            h.ilu_solve("x", "y")
            assert rel2(h.vec_get("y"), c.ilu_solve(x)) < 1e-10
            h.pc_apply("x", "y")
            assert rel2(h.vec_get("y"), c.pc_apply(x)) < 1e-9
        F = c.residual()
        h.residual()
        h.copy_residual_to("b")
        its_h, reason_h, _ = h.fgmres("b", "d")
        d_c, its_c, reason_c, _ = c.fgmres(F)
        assert reason_h == reason_c == 2 and abs(its_h - its_c) <= 1, (name, its_h, its_c)
        its[name] = its_h
        if name == "whole":
            ms = h.time_kernel(1, 10)
            print("whole-slab ILU(0) sweep on C4: %.3f ms per application" % ms)
        h.close()
    print("C4 FGMRES iterations: default tiles %d, one block %d" % (its["tiles"], its["whole"]))
    assert its["whole"] <= its["tiles"]


def _c5_slab_spec():
    """One of the 8 slabs of BASELINE config 5: 240x110x340 cells of 1/4 SPE10 size, 21+21 'large' wells + heaters."""
    import bench
    params, geo, case, cls, kw = bench.build_case("c5slab")
    from thermalporous_amd.problem import build_spec
    spec = build_spec(geo, case, params, 2)
    return spec, cases.uniform_state(spec, params.p_ref, params.T_prod, params.S_o)


def test_c5_slab_properties_9M_cells():
    """What each GPU owns in config 5 (240x110x340 = 8 976 000 cells): flux cancellation, linearity of every
    preconditioner stage, true residual of the FGMRES solution, and one Newton solve of the ramp."""
    from thermalporous_amd.engine import HipEngine
    spec, u0 = _c5_slab_spec()
    assert tuple(spec["n"]) == (340, 110, 240) or sorted(spec["n"]) == [110, 240, 340]
    nosrc = dict(spec)
    nosrc["sources"] = None
    h = HipEngine(nosrc, dict(pc="cptr", ksp_rtol=1e-8))
    u = cases.perturbed_state(spec, seed=1, amp=0.05)
    # cells are 1/4 of the SPE10 size in every direction: |E|/dt shrinks 64x, the face couplings 4x -- dt = 3 s here is
    # the regime of dt = 50 s on config 4 (tests/test_gpu_model.py)
    h.set_old(u)
    h.set_state(u)
    h.set_dt(3.0)
    Rflux = h.residual()                      # u == u_old: pure face fluxes, every field sums to zero
    for f in range(3):
        assert abs(Rflux[f].sum()) <= 1e-9*np.abs(Rflux[f]).sum()
    del Rflux
    h.set_old(u0)
    h.set_state(u)
    h._ck(h.lib.tp_jacobian(h.ctx))           # (no 4.5 GB export of the Jacobian)
    h.pc_setup()
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(u.shape), rng.standard_normal(u.shape)
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
    del mx, my
    h.residual()
    h.copy_residual_to("b")
    its, reason, rn = h.fgmres("b", "d")
    assert reason == 2 and its < 150
    h.spmv("d", "Jd")
    b = h.vec_get("b")
    assert np.linalg.norm(h.vec_get("Jd") - b) <= 1.5e-8*np.linalg.norm(b)
    h.close()
    # with the 42 wells (rate scaled with the cell volume, bench.py): the first time step, driven like the reference's
    # time loop drives it (thermalmodel.py:162-181: dt = maxdt*2^-10, halved while the Newton solve diverges)
    h = HipEngine(spec, dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25))
    dt = 0.1*86400.0/1024.0
    for attempt in range(6):
        h.set_state(u0)
        h.set_old(None)
        h.set_dt(dt)
        r = h.newton_solve()
        if r["reason"] > 0:
            break
        dt *= 0.5
    assert r["reason"] > 0 and 0 < r["nits"] <= 25, (attempt, dt, r)
    smin, smax = h.saturation_range()
    assert -1e-6 <= smin and smax <= 1.0 + 1e-6
    h.close()


def test_c5_shaped_two_slabs_in_process():
    """The 2-slab algorithm (RCCL call sequence, in-process copies) on a C5-shaped box that fits one GPU twice:
    120x56x84 cells of 1/4 SPE10 size cut along x, distributed top AMG levels (amg_gather_cells below the grid size)
    -- same Newton counts and state as the 1-slab run."""
    from test_gpu_slabs import run_slabs
    from thermalporous_amd.engine import HipEngine
    import bench
    from thermalporous_amd.problem import build_spec
    params, geo, case, cls, kw = bench.build_case("c5slab", Nxyz=(120, 56, 84))
    spec = build_spec(geo, case, params, 2)
    u0 = cases.uniform_state(spec, params.p_ref, params.T_prod, params.S_o)
    opts = dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, amg_gather_cells=100000)
    dts = [0.1*86400.0/8192.0, 0.2*86400.0/8192.0]       # (where the ramp of this refined case lands: ~1 s, see bench c5slab)
    h = HipEngine(spec, opts)
    h.set_state(u0)
    ref = []
    for dt in dts:
        h.set_old(None)
        h.set_dt(dt)
        ref.append(h.newton_solve())
    u1 = h.get_state()
    h.close()
    infos, u2 = run_slabs(spec, opts, u0, dts, 2)
    for a, b in zip(ref, infos):
        assert a["reason"] > 0 and b["reason"] == a["reason"] and b["nits"] == a["nits"]
        assert abs(b["lits"] - a["lits"]) <= max(2, 0.1*a["lits"])       # ILU tiles restart at the slab boundary
    assert rel2(u2[0], u1[0]) < 1e-8 and rel2(u2[1], u1[1]) < 1e-8 and np.abs(u2[2] - u1[2]).max() < 1e-8


def test_c5_full_box_eight_slabs_in_process():
    """BASELINE config 5 AS SPECIFIED: the whole 240x880x340 box (71.8 M cells, 215 M unknowns) cut into 8 slabs of 110
    planes, all eight driven through the library's in-process slab group on ONE 288 GB GPU -- the call sequence and buffer
    arithmetic of the 8-GPU RCCL run (halo exchanges, 8-way distributed top AMG levels, the gathered replicated tail,
    batched all-reduces) at the size `bench.py --config c5 --gpus 8` runs.  Size-independent properties: the pure face
    fluxes cancel over the WHOLE box (across the seven slab interfaces), the preconditioner is linear, the FGMRES
    solution has a true residual below the tolerance, the first time step converges with identical counts on every
    slab.  Memory: ~21 GB per slab without the Krylov basis; ksp_restart = 10 keeps the eight bases at 4.8 GB each
    (the default restart of 200 grows the basis on demand and is what an 8-GPU run uses: one slab per 288 GB)."""
    import ctypes as C
    import sys
    import threading
    import time
    import bench
    from thermalporous_amd import engine as E
    from thermalporous_amd.problem import build_spec
    t_start = time.time()
    import faulthandler
    faulthandler.dump_traceback_later(600, exit=False, file=sys.stderr)      # a stuck slab thread shows where

    def say(msg):           # progress on stderr: minutes pass between the phases of this test
        print("[c5 full box %6.1f s] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)
    params, geo, case, cls, kw = bench.build_case("c5")
    say("fields and wells built")
    spec = build_spec(geo, case, params, 2)
    assert sorted(spec["n"]) == [240, 340, 880] and spec["n"][2] == 880
    del geo, case
    u0 = cases.uniform_state(spec, params.p_ref, params.T_prod, params.S_o)
    u = cases.perturbed_state(spec, seed=1, amp=0.05)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(u.shape), rng.standard_normal(u.shape)
    z = 2.0*x - 3.0*y
    nranks = 8
    lib = E.load_library()
    group = C.c_void_p()
    assert lib.tp_local_group_create(nranks, C.byref(group)) == 0
    opts = dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, ksp_restart=10, ksp_max_it=400)
    out, err = [None]*nranks, []

    def worker(rank):
        try:
            nosrc = dict(spec)
            nosrc["sources"] = None
            h = E.HipEngine(nosrc, opts, rank=rank, nranks=nranks, local_group=group)
            res = {}
            h.set_old(u)
            h.set_state(u)
            h.set_dt(3.0)
            R = h.residual()                                  # u == u_old: pure face fluxes
            res["flux_sum"] = [float(R[f].sum()) for f in range(3)]
            res["flux_abs"] = [float(np.abs(R[f]).sum()) for f in range(3)]
            del R
            if rank == 0:
                say("flux residual done")
            h.set_old(u0)
            h.set_state(u)
            h._ck(h.lib.tp_jacobian(h.ctx))
            h.pc_setup()
            res["layout"] = h.amg_layout(0)
            if rank == 0:
                say("Jacobian + pc_setup done, layout %r" % (res["layout"],))
            for name, v in (("mx", x), ("my", y), ("mz", z)):
                h.vec_set("x", v)
                h.pc_apply("x", name)
            mx, my, mz = h.vec_get("mx"), h.vec_get("my"), h.vec_get("mz")
            d = mz - (2.0*mx - 3.0*my)
            res["lin"] = (float((d*d).sum()), float(((2.0*mx - 3.0*my)**2).sum()))
            del mx, my, mz, d
            if rank == 0:
                say("pc_apply linearity done")
            h.residual()
            h.copy_residual_to("b")
            res["fgmres"] = h.fgmres("b", "d")
            h.spmv("d", "Jd")
            b = h.vec_get("b")
            r = h.vec_get("Jd") - b
            res["true_res"] = (float((r*r).sum()), float((b*b).sum()))
            del b, r
            if rank == 0:
                say("fgmres done: %r" % (res["fgmres"],))
            h.close()
            # the first time step of the reference's time loop with the 42 wells (dt = maxdt*2^-10, halved on divergence)
            h = E.HipEngine(spec, opts, rank=rank, nranks=nranks, local_group=group)
            dt = 0.1*86400.0/8192.0            # (three halvings below the ramp's first dt: where the one-slab probe lands too)
            for attempt in range(4):
                h.set_state(u0)
                h.set_old(None)
                h.set_dt(dt)
                info = h.newton_solve()
                if rank == 0:
                    say("newton attempt %d dt %.3g: %r" % (attempt, dt, info))
                if info["reason"] > 0:             # (identical on every slab: the norms are all-reduced)
                    break
                dt *= 0.5
            res["newton"] = (info["nits"], info["lits"], info["reason"], attempt)
            res["srange"] = h.saturation_range()
            h.close()
            out[rank] = res
        except Exception as e:      # noqa: BLE001
            err.append((rank, repr(e)))
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    faulthandler.cancel_dump_traceback_later()
    assert not any(t.is_alive() for t in ts), "slab worker hung"
    lib.tp_local_group_destroy(group)
    assert not err, err
    # 8-way distributed top levels: 71.8 M -> ... -> the first level of at most amg_gather_cells = 2 M cells is gathered
    nd, sched = out[0]["layout"]
    assert nd >= 5 and all(o["layout"] == out[0]["layout"] for o in out), out[0]["layout"]
    for f in range(3):              # fluxes cancel over the whole box, slab interfaces included
        assert abs(sum(o["flux_sum"][f] for o in out)) <= 1e-9*sum(o["flux_abs"][f] for o in out)
    assert sum(o["lin"][0] for o in out) <= (1e-9)**2*sum(o["lin"][1] for o in out)
    its, reason, _ = out[0]["fgmres"]
    assert reason == 2 and all(o["fgmres"][:2] == (its, reason) for o in out), [o["fgmres"] for o in out]
    assert sum(o["true_res"][0] for o in out) <= (1.5e-8)**2*sum(o["true_res"][1] for o in out)
    assert all(o["newton"] == out[0]["newton"] for o in out)
    nits, lits, nreason, attempt = out[0]["newton"]
    assert nreason > 0 and 0 < nits <= 25, out[0]["newton"]
    assert min(o["srange"][0] for o in out) >= -1e-6 and max(o["srange"][1] for o in out) <= 1.0 + 1e-6
    print("c5 full box, 8 slabs: dist levels %d, fgmres(10) %d its, first step: %d Newton / %d Krylov its (dt halvings %d)"
          % (nd, its, nits, lits, attempt))
