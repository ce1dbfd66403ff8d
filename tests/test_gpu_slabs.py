"""The N-slab (multi-GPU) algorithm validated on ONE GPU: N HipEngine contexts driven by N host threads
exchange halos / gather the stage-1 system / reduce dot products through the library's in-process slab
group (tp_comm_init_local) -- the same call sequence and buffer arithmetic as the RCCL path, with
device-to-device copies instead of ncclSend/Recv/Broadcast/AllReduce.  Checked against the 1-slab HIP
run and the oracle's N-slab emulation."""
import ctypes as C
import threading

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def rel2(a, b):
    return np.linalg.norm((a - b).ravel())/max(np.linalg.norm(b.ravel()), 1e-300)


def run_slabs(spec, opts, u0, dts, nranks):
    from thermalporous_amd import engine as E
    lib = E.load_library()
    group = C.c_void_p()
    assert lib.tp_local_group_create(nranks, C.byref(group)) == 0
    out = [None]*nranks
    err = []

    def worker(rank):
        try:
            h = E.HipEngine(spec, opts, rank=rank, nranks=nranks, local_group=group)
            h.set_state(u0)
            infos = []
            for dt in dts:
                h.set_old(None)
                h.set_dt(dt)
                infos.append(h.newton_solve())
            rng = h.saturation_range() if h.b == 3 else None
            out[rank] = (infos, h.get_state(), rng)
            h.close()
        except Exception as e:      # noqa: BLE001
            err.append((rank, repr(e)))
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "slab worker hung"
    lib.tp_local_group_destroy(group)
    assert not err, err
    infos = out[0][0]
    for r in range(1, nranks):      # every rank must report identical solver statistics
        assert [(i["nits"], i["lits"], i["reason"]) for i in out[r][0]] == [(i["nits"], i["lits"], i["reason"]) for i in infos]
    state = np.concatenate([o[1] for o in out], axis=1)
    return infos, state


def run_linear_stage(spec, opts, u0, u, dt, xs, nranks, vcycles=True):
    """pc_setup on every slab, then the pressure V-cycle, (S~ V-cycle,) stage 1 and the whole pc_apply of the
    global vector xs; returns the assembled global results and the AMG layout of rank 0."""
    from thermalporous_amd import engine as E
    lib = E.load_library()
    group = C.c_void_p()
    if nranks > 1:
        assert lib.tp_local_group_create(nranks, C.byref(group)) == 0
    out = [None]*nranks
    err = []

    def worker(rank):
        try:
            h = E.HipEngine(spec, opts, rank=rank, nranks=nranks, local_group=group if nranks > 1 else None)
            h.set_old(u0)
            h.set_dt(dt)
            h.set_state(u)
            h.jacobian()
            h.pc_setup()
            h.vec_set("x", xs)
            res = {}
            # (the system-AMG presets have no scalar pressure hierarchy; tp_amg_vcycle is not exported for replicated hierarchies)
            if vcycles and opts["pc"] != "cptramg":
                h.amg_vcycle(0, "x", 0, "v0", 0)
                res["v0"] = h.vec_get("v0")[0]
            if vcycles and opts["pc"] == "cptr":
                h.amg_vcycle(1, "x", 1, "v1", 1)
                res["v1"] = h.vec_get("v1")[1]
            h.vec_set("x", xs)
            h.stage1_apply("x", "s1")
            res["s1"] = h.vec_get("s1")
            h.vec_set("x", xs)
            h.pc_apply("x", "pc")
            res["pc"] = h.vec_get("pc")
            out[rank] = (res, h.amg_layout(2 if opts["pc"] == "cptramg" else 0), h.amg_trunc(1)[0] if opts["pc"] == "cptr" else None)
            h.close()
        except Exception as e:      # noqa: BLE001
            err.append((rank, repr(e)))
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "slab worker hung"
    if nranks > 1:
        lib.tp_local_group_destroy(group)
    assert not err, err
    res = {k: np.concatenate([o[0][k] for o in out], axis=-3) for k in out[0][0]}
    assert len({o[2] for o in out}) == 1        # every rank took the same relaxation-only truncation decision
    return res, out[0][1], out[0][2]


DIST_AMG_CASES = [
    # (name, grid kw, opts): amg_gather_cells = 0 keeps every level with >= 2 planes per slab distributed
    ("2ph_cptr", dict(Nx=8, Ny=21, Nz=7, nphase=2), dict(pc="cptr", amg_gather_cells=0)),
    ("2ph_cptr_fp32", dict(Nx=6, Ny=26, Nz=5, nphase=2), dict(pc="cptr", amg_gather_cells=0, amg_single=True)),
    ("2ph_cptr_v11", dict(Nx=5, Ny=33, Nz=4, nphase=2), dict(pc="cptr", amg_gather_cells=0, amg_nu=1, amg_full_levels=99)),
    ("2ph_cptr_v33", dict(Nx=5, Ny=24, Nz=6, nphase=2), dict(pc="cptr", amg_gather_cells=0, amg_nu=3, amg_full_levels=2)),
    ("2ph_cptr_partial", dict(Nx=8, Ny=21, Nz=7, nphase=2), dict(pc="cptr", amg_gather_cells=400)),
    ("1ph_cprQI", dict(Nx=6, Ny=17, Nz=5, nphase=1), dict(pc="cpr", decoup="QI", amg_gather_cells=0)),
    # small dt: S~ is strongly diagonally dominant and its DISTRIBUTED level 0 ends the cycle (amg_dom_tau; the ranks
    # agree on the decision through a max-all-reduce of the per-slab dominance ratios)
    ("2ph_cptr_trunc", dict(Nx=8, Ny=21, Nz=7, nphase=2), dict(pc="cptr", amg_gather_cells=0, _dt=86.4)),
]


@pytest.mark.parametrize("name,kw,opts", DIST_AMG_CASES, ids=[c[0] for c in DIST_AMG_CASES])
@pytest.mark.parametrize("nranks", [2, 3])
def test_distributed_amg_levels_equal_the_single_slab_hierarchy(name, kw, opts, nranks):
    """The slab-distributed top AMG levels are the same algebra as the single-slab hierarchy (C points = even
    GLOBAL planes, Jacobi smoothing, non-Galerkin coarse operators): V-cycles and stage 1 agree to round-off;
    the full pc_apply differs only through the per-slab ILU tiles and is checked against the N-slab oracle."""
    from oracle.engine import OracleEngine
    spec, u0, *_ = cases.c4_spe10_3d(**kw)
    u = cases.perturbed_state(spec, seed=5, amp=0.2)
    xs = np.random.default_rng(11).standard_normal(u.shape)
    opts = dict(opts)
    dt = opts.pop("_dt", 3000.0)
    one, lay1, t1 = run_linear_stage(spec, opts, u0, u, dt, xs, 1)
    many, layn, tn = run_linear_stage(spec, opts, u0, u, dt, xs, nranks)
    assert t1 == tn                                                    # same cycle shape on one slab and on N
    if name.endswith("trunc"):
        assert 0 <= tn < layn[0], (tn, layn)                           # ... ending on a slab-distributed level
    assert lay1[0] == 0 and layn[0] >= 1 and lay1[1] == layn[1]        # same schedule, top levels distributed
    assert 2 in layn[1][:layn[0]] or name.endswith("partial")          # a distributed level coarsens the slab axis
    tol = 2e-6 if opts.get("amg_single") else 1e-11
    for k in ("v0", "v1", "s1"):
        if k in one:
            assert rel2(many[k], one[k]) < tol, k
    o = OracleEngine(spec, dict(opts, nslabs=nranks))
    o.set_old(u0)
    o.set_dt(dt)
    o.set_state(u)
    schur = opts["pc"] == "cptr"
    jo = o.jacobian(want_schur=schur)
    J, Sm = jo if schur else (jo, None)
    o.pc.setup(J, Sm)
    assert rel2(many["pc"], o.pc.apply(xs)) < (2e-6 if opts.get("amg_single") else 1e-9)


def test_random_boxes_on_random_slab_counts():
    """Seeded fuzz of the N-slab linear stage: random boxes, 2-5 slabs (ragged slab heights), random gather thresholds (replicated,
    partly and fully distributed hierarchies), presets and cycle shapes.  Stage 1 equals the single slab's to round-off, the whole
    pc_apply equals the N-slab oracle's."""
    from oracle.engine import OracleEngine
    rng = np.random.default_rng(510)
    kinds = [dict(pc="cptr"), dict(pc="cptr", decoup="QI"), dict(pc="cpr", decoup="QI"), dict(pc="cptramg", decoup="QI"),
             dict(pc="cptr", ilu_whole=True, ilu_tile=(4, 3, 3))]
    for it in range(10):
        nranks = int(rng.integers(2, 6))
        Nx, Nz = int(rng.integers(2, 9)), int(rng.integers(2, 8))
        Ny = int(rng.integers(max(2*nranks, 9), 34))                # the longest direction carries the slabs
        opts = dict(kinds[int(rng.integers(0, len(kinds)))])
        nphase = 2 if opts["pc"] != "cpr" or rng.integers(0, 2) else 1
        opts.update(amg_gather_cells=int(rng.choice([0, 100, 400, 2000000])), amg_full_levels=int(rng.integers(0, 4)),
                    amg_mid_skip=bool(rng.integers(0, 2)), amg_nu=int(rng.integers(1, 3)))
        tag = (it, nranks, Nx, Ny, Nz, nphase, opts)
        spec, u0, *_ = cases.c4_spe10_3d(Nx=Nx, Ny=Ny, Nz=Nz, nphase=nphase)
        u = cases.perturbed_state(spec, seed=70 + it, amp=0.2)
        xs = rng.standard_normal(u.shape)
        dt = 3000.0
        one, _, _ = run_linear_stage(spec, opts, u0, u, dt, xs, 1, vcycles=False)
        many, _, _ = run_linear_stage(spec, opts, u0, u, dt, xs, nranks, vcycles=False)
        assert rel2(many["s1"], one["s1"]) < 1e-10, tag
        o = OracleEngine(spec, dict(opts, nslabs=nranks))
        o.set_old(u0)
        o.set_dt(dt)
        o.set_state(u)
        schur = opts["pc"] == "cptr"
        jo = o.jacobian(want_schur=schur)
        J, Sm = jo if schur else (jo, None)
        o.pc.setup(J, Sm)
        assert rel2(many["pc"], o.pc.apply(xs)) < 1e-9, tag


SLAB_CASES = [
    ("2ph_cptr", cases.c4_spe10_3d, dict(Nx=8, Ny=21, Nz=7, nphase=2), dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25), [40.0, 80.0]),
    ("1ph_cprTI", cases.c4_spe10_3d, dict(Nx=6, Ny=17, Nz=5, nphase=1), dict(pc="cpr", decoup="TI", ksp_rtol=1e-8), [400.0]),
    ("2ph_cprQI", cases.c4_spe10_3d, dict(Nx=7, Ny=16, Nz=6, nphase=2), dict(pc="cpr", decoup="QI", ksp_rtol=1e-8, snes_max_it=25), [40.0]),
    # the same with the top AMG levels distributed over the slabs instead of replicated
    ("2ph_cptr_distamg", cases.c4_spe10_3d, dict(Nx=8, Ny=21, Nz=7, nphase=2),
     dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, amg_gather_cells=0), [40.0, 80.0]),
    ("1ph_cprTI_distamg", cases.c4_spe10_3d, dict(Nx=6, Ny=17, Nz=5, nphase=1),
     dict(pc="cpr", decoup="TI", ksp_rtol=1e-8, amg_gather_cells=100), [400.0]),
    # one bjacobi block per slab (PETSc's default bjacobi): whole-slab ILU(0) on every rank
    ("2ph_cptr_whole", cases.c4_spe10_3d, dict(Nx=8, Ny=21, Nz=7, nphase=2),
     dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, ilu_whole=True, ilu_tile=(4, 3, 3)), [40.0, 80.0]),
]


@pytest.mark.parametrize("name,builder,kw,opts,dts", SLAB_CASES, ids=[c[0] for c in SLAB_CASES])
@pytest.mark.parametrize("nranks", [2, 3])
def test_slab_group_matches_single_slab_and_oracle(name, builder, kw, opts, dts, nranks):
    from oracle.engine import OracleEngine
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = builder(**kw)
    # single slab on the GPU
    h = HipEngine(spec, opts)
    h.set_state(u0)
    ref = []
    for dt in dts:
        h.set_old(None)
        h.set_dt(dt)
        ref.append(h.newton_solve())
    u_ref = h.get_state()
    h.close()
    assert all(r["reason"] > 0 for r in ref)
    # N slabs on the same GPU
    infos, u_n = run_slabs(spec, opts, u0, dts, nranks)
    # oracle emulation of the same N-slab algorithm
    o = OracleEngine(spec, dict(opts, nslabs=nranks))
    o.set_state(u0)
    orc = []
    for dt in dts:
        o.set_old(o.get_state())
        o.set_dt(dt)
        orc.append(o.newton_solve())
    for i_n, i_1, i_o in zip(infos, ref, orc):
        assert i_n["reason"] == i_1["reason"] == i_o["reason"]
        assert i_n["nits"] == i_1["nits"] == i_o["nits"]
        assert abs(i_n["lits"] - i_o["lits"]) <= 2                      # same algorithm as the emulation
        assert abs(i_n["lits"] - i_1["lits"]) <= max(3, 0.2*i_1["lits"])  # ILU tiling differs from 1 slab
    for f in range(u_ref.shape[0]):
        assert rel2(u_n[f], u_ref[f]) < 1e-7
        assert rel2(u_n[f], o.get_state()[f]) < 1e-7


@pytest.mark.parametrize("gather", [2000000, 0], ids=["replicated", "distributed"])
def test_rccl_calls_on_a_one_rank_communicator(gather):
    """RCCL refuses two ranks on one GPU, so the N-rank exchange cannot be rehearsed on this box; a ONE-rank
    communicator still runs every RCCL entry point the slab path uses (ncclCommInitRank, grouped send/recv with
    no neighbours, grouped in-place ncclBroadcast gathers, ncclAllReduce on the compute stream) and must change
    nothing in the results."""
    from thermalporous_amd.engine import HipEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=8, Ny=21, Nz=7, nphase=2)
    opts = dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25, amg_gather_cells=gather)
    res = []
    for boot in (None, lambda make_id: make_id()):
        h = HipEngine(spec, opts, rank=0, nranks=1, comm_bootstrap=boot)
        h.set_state(u0)
        infos = []
        for dt in (40.0, 80.0):
            h.set_old(None)
            h.set_dt(dt)
            infos.append(h.newton_solve())
        res.append(([(i["nits"], i["lits"], i["reason"]) for i in infos], h.get_state(), h.amg_layout(0)[0]))
        h.close()
    assert res[0][0] == res[1][0] and all(r[2] > 0 for r in res[0][0])
    assert res[0][2] == 0 and (res[1][2] > 0) == (gather == 0)
    for f in range(3):
        assert rel2(res[1][1][f], res[0][1][f]) < 1e-9


def _pc_sequence_on_slabs(spec, u0, u, dt, xs, nranks, seq):
    """On every slab of an in-process group: for each options dict of `seq` in turn, set_options -> jacobian ->
    pc_setup -> pc_apply(xs).  Returns the global results of every stage of the sequence."""
    from thermalporous_amd import engine as E
    lib = E.load_library()
    group = C.c_void_p()
    assert lib.tp_local_group_create(nranks, C.byref(group)) == 0
    out = [None]*nranks
    err = []

    def worker(rank):
        try:
            h = E.HipEngine(spec, seq[0], rank=rank, nranks=nranks, local_group=group)
            h.set_old(u0)
            h.set_dt(dt)
            h.set_state(u)
            res = []
            for k, o in enumerate(seq):
                if k:
                    h.set_options(**o)
                h.jacobian()
                h.pc_setup()
                h.vec_set("x", xs)
                h.pc_apply("x", "y")
                res.append(h.vec_get("y"))
            out[rank] = res
            h.close()
        except Exception as e:      # noqa: BLE001
            err.append((rank, repr(e)))
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "slab worker hung"
    lib.tp_local_group_destroy(group)
    assert not err, err
    return [np.concatenate([o[k] for o in out], axis=-3) for k in range(len(seq))]


@pytest.mark.parametrize("seq_name", ["cptr_cptramg_cptr", "cpr_cptr"])
def test_switching_pc_kinds_on_a_live_slab_group(seq_name):
    """tp_set_options on a live multi-slab context: the global-grid scratch buffers (gvec, gA00, gA01, gA10, gSm,
    gAt) are each sized from their own needs, so cptr -> cptramg -> cptr does not leave gvec with 4 planes where stage 1
    writes 6, and cpr -> cptr does not leave gA01/gA10/gSm unallocated.  Every stage must equal a FRESH context's result."""
    spec, u0, *_ = cases.c4_spe10_3d(Nx=8, Ny=21, Nz=7, nphase=2)
    u = cases.perturbed_state(spec, seed=5, amp=0.2)
    xs = np.random.default_rng(11).standard_normal(u.shape)
    base = dict(ksp_rtol=1e-8, amg_gather_cells=2000000)          # replicated hierarchies: the global-grid scratch is in use
    seqs = {"cptr_cptramg_cptr": [dict(base, pc="cptr"), dict(base, pc="cptramg", decoup="QI"), dict(base, pc="cptr", decoup="No")],
            "cpr_cptr": [dict(base, pc="cpr", decoup="QI"), dict(base, pc="cptr", decoup="No")]}
    seq = seqs[seq_name]
    got = _pc_sequence_on_slabs(spec, u0, u, 3000.0, xs, 2, seq)
    for k, o in enumerate(seq):
        fresh = _pc_sequence_on_slabs(spec, u0, u, 3000.0, xs, 2, [o])[0]
        assert rel2(got[k], fresh) < 1e-12, (seq_name, k)


@pytest.mark.parametrize("gather", [2000000, 0], ids=["replicated", "distributed"])
def test_pc_apply_program_replay_on_slabs(gather):
    """Multi-slab pc_apply runs as a recorded PROGRAM: hipGraph segments between the exchanges, the exchanges as host
    closures (tp_common.hpp).  The recording pass must already produce the result, a replay on the same vector pair must
    reproduce it bit for bit, and a replay with other CONTENTS in the same buffers must follow them (linearity)."""
    from thermalporous_amd import engine as E
    spec, u0, *_ = cases.c4_spe10_3d(Nx=8, Ny=21, Nz=7, nphase=2)
    u = cases.perturbed_state(spec, seed=5, amp=0.2)
    xs = np.random.default_rng(11).standard_normal(u.shape)
    opts = dict(pc="cptr", ksp_rtol=1e-8, amg_gather_cells=gather)
    nranks = 3
    lib = E.load_library()
    group = C.c_void_p()
    assert lib.tp_local_group_create(nranks, C.byref(group)) == 0
    out, err = [None]*nranks, []

    def worker(rank):
        try:
            h = E.HipEngine(spec, opts, rank=rank, nranks=nranks, local_group=group)
            h.set_old(u0)
            h.set_dt(3000.0)
            h.set_state(u)
            h.jacobian()
            h.pc_setup()
            res = []
            for fac in (1.0, 1.0, -2.5):          # record, replay, replay with other contents
                h.vec_set("x", fac*xs)
                h.pc_apply("x", "y")
                res.append(h.vec_get("y"))
            out[rank] = res
            h.close()
        except Exception as e:      # noqa: BLE001
            err.append((rank, repr(e)))
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "slab worker hung"
    lib.tp_local_group_destroy(group)
    assert not err, err
    y1, y2, y3 = (np.concatenate([o[k] for o in out], axis=-3) for k in range(3))
    assert np.array_equal(y1, y2)
    assert rel2(y3, -2.5*y1) < 1e-13
    # and it is the N-slab oracle's preconditioner
    from oracle.engine import OracleEngine
    o = OracleEngine(spec, dict(opts, nslabs=nranks))
    o.set_old(u0)
    o.set_dt(3000.0)
    o.set_state(u)
    J, Sm = o.jacobian(want_schur=True)
    o.pc.setup(J, Sm)
    assert rel2(y1, o.pc.apply(xs)) < 1e-9


@pytest.mark.parametrize("nranks", [2, 3])
def test_selfp_on_slabs_equals_single_slab(nranks):
    """pc_fieldsplit_selfp (singlephase.py:322-330) on several slabs: Sp = A11 - A10 diag(A00)^-1 A01 of a cell next to a slab
    boundary reads its neighbour's diag(A00) and A01 row from the exchanged halo rows of the Jacobian, the exact-Sp sweep
    exchanges x and u.  No ILU stage in this preset, so the N-slab preconditioner IS the single-slab one (to round-off)
    and the oracle's."""
    from oracle.engine import OracleEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=6, Ny=23, Nz=5, nphase=1)
    u = cases.perturbed_state(spec, seed=5, amp=0.2)
    xs = np.random.default_rng(11).standard_normal(u.shape)
    opts = dict(pc="fieldsplit_cd", schur_selfp=True)
    one, lay1, _ = run_linear_stage(spec, opts, u0, u, 3000.0, xs, 1)
    many, layn, _ = run_linear_stage(spec, opts, u0, u, 3000.0, xs, nranks)
    assert lay1[0] == 0 and layn[0] >= 1            # (selfp keeps the top levels on the slabs whatever amg_gather_cells says)
    for k in ("v0", "s1", "pc"):
        assert rel2(many[k], one[k]) < 1e-11, k
    o = OracleEngine(spec, opts)
    o.set_old(u0)
    o.set_dt(3000.0)
    o.set_state(u)
    J, Sm = o.jacobian(want_schur=True)
    o.pc.setup(J, Sm)
    assert rel2(many["pc"], o.pc.apply(xs)) < 1e-9


@pytest.mark.parametrize("decoup", ["No", "QI"])
@pytest.mark.parametrize("nranks", [2, 3])
def test_system_amg_on_slabs_equals_single_slab(decoup, nranks):
    """pc_cptramg (twophase.py:552-566) with its 2x2-block hierarchy DISTRIBUTED over the slabs (round 3: C points = even global
    planes, halo exchanges of the (p,T) vectors, weights / inverse diagonal blocks / slab-axis operator rows exchanged per
    set-up, first small level gathered in place): stage 1 equals the single-slab hierarchy to round-off; the whole
    preconditioner equals the N-slab oracle's (ILU tiles restart per slab)."""
    from oracle.engine import OracleEngine
    spec, u0, *_ = cases.c4_spe10_3d(Nx=8, Ny=21, Nz=7, nphase=2)
    u = cases.perturbed_state(spec, seed=5, amp=0.2)
    xs = np.random.default_rng(11).standard_normal(u.shape)
    opts = dict(pc="cptramg", decoup=decoup, amg_gather_cells=0)

    def stage(n):
        from thermalporous_amd import engine as E
        lib = E.load_library()
        group = C.c_void_p()
        if n > 1:
            assert lib.tp_local_group_create(n, C.byref(group)) == 0
        out, err = [None]*n, []

        def worker(rank):
            try:
                h = E.HipEngine(spec, opts, rank=rank, nranks=n, local_group=group if n > 1 else None)
                h.set_old(u0)
                h.set_dt(3000.0)
                h.set_state(u)
                h.jacobian()
                h.pc_setup()
                h.vec_set("x", xs)
                h.stage1_apply("x", "s1")
                s1 = h.vec_get("s1")
                h.vec_set("x", xs)
                h.pc_apply("x", "pc")
                out[rank] = (s1, h.vec_get("pc"), h.amg_layout(2))
                h.close()
            except Exception as e:      # noqa: BLE001
                err.append((rank, repr(e)))
        ts = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in ts), "slab worker hung"
        if n > 1:
            lib.tp_local_group_destroy(group)
        assert not err, err
        return [np.concatenate([o[k] for o in out], axis=-3) for k in range(2)] + [out[0][2]]
    s1_one, _, lay1 = stage(1)
    s1_n, pc_n, layn = stage(nranks)
    assert lay1[0] == 0 and layn[0] >= 2 and 2 in layn[1][:layn[0]], (lay1, layn)     # distributed, incl. a slab-axis level
    assert rel2(s1_n, s1_one) < 1e-11
    o = OracleEngine(spec, dict(opts, nslabs=nranks))
    o.set_old(u0)
    o.set_dt(3000.0)
    o.set_state(u)
    o.pc.setup(o.jacobian())
    assert rel2(pc_n, o.pc.apply(xs)) < 1e-9
