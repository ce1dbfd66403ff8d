#!/usr/bin/env python3
"""bench.py -- Newton steps/s (+ FGMRES its/s) of the hot path; default = BASELINE config 4:
two-phase 3-D SPE10-like 60x220x85 box, wells + heaters, pc_cptr (CPTR: fieldsplit-Schur stage 1 with
AMG V-cycles on App and S~, block-Jacobi block-ILU(0) stage 2) inside FGMRES inside Newton.

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5|c5slab]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one time step of the reference's time loop = one ``solver.solve()`` (one Newton solve with
its linear solves) + the loop's dt policy (thermalporous_amd/thermalmodel.py:step).  All inputs are
synthetic (the SPE10 .dat files are not shipped with the reference) and resident in HBM before the
timed region.

Time-stepping regime ("time steps of 0.1 day" on config 4).  From the uniform initial state a
cold 0.1-day step does not converge with the reference's `basic` line search, so the run first SPINS UP with
the reference's own dt ramp (dt = maxdt*2^-10, growing by the SPE10 rule of thermalmodel.py:337-345) until dt
has reached maxdt -- untimed, not counted in W -- then does W warm-up steps and K timed steps, all at
dt = maxdt unless the reference's failure policy cuts dt (reported: dt range, failed solves).  The rate seen
during the spin-up ramp is reported as a secondary field (``ramp``).

N > 1: the fixed box is cut into N slabs along the internal slab axis ("strong" scaling).
Rank 0 prints ONE JSON line: the metric, a ``roofline`` object for the dominant single kernel of a Krylov
iteration (the block-ILU(0) solve) with the block SpMV and the other streaming kernels beside it, and the
``cpu_baseline`` (oracle/cport: the C++/OpenMP restatement of the same algorithm, timed on this box's host
cores from the SAME state and dt as the first timed step; never the thing measured).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SPMV_BYTES_PER_CELL = {(3, 7): 584, (3, 5): 432, (2, 7): 292, (2, 5): 216}
# algorithmic bytes per cell of the other streaming kernels, keyed like SPMV_BYTES_PER_CELL:
#   ILU(0) solve = SpMV;  assembly = 3b*8 + 8 + d*8 + 8 + s*b*b*8;  ILU factor = 2*s*b*b*8 + idx
ASM_BYTES_PER_CELL = {(3, 7): 616, (3, 5): 464, (2, 7): 296, (2, 5): 232}
FACTOR_BYTES_PER_CELL = {(3, 7): 1040, (3, 5): 744, (2, 7): 480, (2, 5): 344}

CONFIGS = {
    # name: (description, default grid)
    "c1": ("BASELINE config 1: single-phase 2D homogeneous NxN (tests/test_homo_wells.py), const-rate wells, pc_cpr, "
           "dt 1 day", (400, 400, 1)),
    "c2": ("BASELINE config 2: single-phase 2D SPE10-like 60x220 layer, Peaceman wells, pc_cpr, maxdt 1 day", (60, 220, 1)),
    "c3": ("BASELINE config 3: two-phase 2D SPE10-like 60x220 layer, Peaceman wells, pc_cptr, maxdt 1 day", (60, 220, 1)),
    "c4": ("BASELINE config 4: two-phase 3D SPE10-like 60x220x85, wells+heaters, pc_cptr, maxdt 0.1 day", (60, 220, 85)),
    "c5": ("BASELINE config 5: two-phase 3D 240x880x340 (60x220x85 field upsampled x4, dead cells drawn at the fine "
           "resolution), 21+21 constant-rate 'large' wells (1e-7 m^3/s), pc_cptr, maxdt 0.1 day -- the HBM-fill case for "
           "--gpus 8 (71.8 M cells: does not fit one GPU)", (240, 880, 340)),
    "c5slab": ("BASELINE config 5, ONE of its 8 slabs: two-phase 3D 240x110x340 (60x220x85 field upsampled x4), "
               "21+21 constant-rate 'large' wells (1e-7 m^3/s), pc_cptr, maxdt 0.1 day", (240, 110, 340)),
}


def build_case(name, Nxyz=None):
    """BASELINE configs on synthetic data.  Returns (params, geo, case, model class, model kwargs)."""
    from thermalporous_amd.physicalparameters import PhysicalParameters
    params = PhysicalParameters()
    Nx, Ny, Nz = Nxyz or CONFIGS[name][1]
    if name == "c1":
        from thermalporous_amd.homogeneousgeo import HomogeneousGeo
        from thermalporous_amd.wellcase import WellCase
        from thermalporous_amd.singlephase import SinglePhase
        params.rate = 1e-6          # tests/test_homo_wells.py:10-12 of the reference
        params.T_prod = 320.0
        geo = HomogeneousGeo(Nx, Ny, params, 20.0, 20.0)
        case = WellCase(params, geo, well_case="test0", constant_rate=True)
        return params, geo, case, SinglePhase, dict(maxdt=1.0, small_dt_start=False, solver_parameters="pc_cpr")
    if name in ("c2", "c3"):
        from thermalporous_amd.SPE10model import SPE10Model
        from thermalporous_amd.wellcase import WellCase
        two = name == "c3"
        params.rate = 2e-4 if two else 1e-3      # tests_twophase/test_60x120_wells_default.py:8-9 / tests/..._default.py:8
        if two:
            params.S_o = 0.9
        geo = SPE10Model(Nx, Ny, params)
        L, Ly = geo.Length, geo.Length_y
        case = WellCase(params, geo, prod_points=[[140.0/365.76*L, 210.0/670.56*Ly]],
                        inj_points=[[265.0/365.76*L, 260.0/670.56*Ly]])
        if two:
            from thermalporous_amd.twophase import TwoPhase
            return params, geo, case, TwoPhase, dict(maxdt=1.0, small_dt_start=True, solver_parameters="pc_cptr")
        from thermalporous_amd.singlephase import SinglePhase
        return params, geo, case, SinglePhase, dict(maxdt=1.0, small_dt_start=True, solver_parameters="pc_cpr")
    from thermalporous_amd.SPE10model3D import SPE10Model3D
    from thermalporous_amd.wellheatercase import WellHeaterCase
    from thermalporous_amd.twophase import TwoPhase
    params.rate = 2e-4          # tests_twophase/test_60x120_wells_default.py:8-9 of the reference
    params.S_o = 0.9
    params.T_inj = 373.15
    if name == "c4":
        geo = SPE10Model3D(Nx, Ny, Nz, params)
        L, Ly, Lz = geo.Length, geo.Length_y, geo.Length_z
        # SPE10 well (x,y) positions (wellcase.py:30-36), producer low / injector high in the column
        prod = [[140.0/365.76*L, 210.0/670.56*Ly, 0.2*Lz]]
        inj = [[265.0/365.76*L, 260.0/670.56*Ly, 0.8*Lz]]
        case = WellHeaterCase(params, geo, prod_points=prod, inj_points=inj)
    elif name in ("c5", "c5slab"):
        geo = SPE10Model3D(Nx, Ny, Nz, params, refine=4)      # cells 1/4 of the SPE10 size in every direction
        # 21 + 21 wells on the 'large' pattern (wellcase.py:58-64) driven as the reference's own 3-D runs drive them:
        # constant rate 1e-7 m^3/s (tests_twophase/test3D_homo_wells.py:11,90).  Peaceman wells at config 4's 2e-4 m^3/s
        # are not usable on cells of 0.17 m^3: Newton stalls in a limit cycle of the rate-cap branch at any dt
        params.rate = 1e-7
        from thermalporous_amd.wellcase import WellCase
        prod, inj = WellCase(params, geo, well_case=None).named_points("large")
        # The reference runs this pattern on homogeneous fields, where every cell is live.  On the SPE10-like field 2.5 %
        # of the cells have zero porosity and the permeability spans 8 decades: a constant-rate well in a dead cell has
        # no physical solution path (measured: the time loop collapses dt to 6e-3 s).  Each well is therefore completed
        # in the most permeable porous cell of its 3x3x3 neighbourhood, as a reservoir engineer would.
        prod, inj = ([_live_cell_centre(geo, w) for w in pts] for pts in (prod, inj))
        case = WellCase(params, geo, prod_points=prod, inj_points=inj, constant_rate=True)
    else:
        raise ValueError(name)
    return params, geo, case, TwoPhase, dict(maxdt=0.1, small_dt_start=True, solver_parameters="pc_cptr")


def _live_cell_centre(geo, w):
    """Centre of the cell with the largest phi*K_x among the 27 cells around the point w = [x, y, z]."""
    D = (geo.Dx, geo.Dy, geo.Dz)
    N = (geo.Nx, geo.Ny, geo.Nz)
    i0 = [min(max(int(w[a]/D[a]), 0), N[a] - 1) for a in range(3)]
    best, arg = -1.0, i0
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for dk in (-1, 0, 1):
                i, j, k = i0[0] + di, i0[1] + dj, i0[2] + dk
                if not (0 <= i < N[0] and 0 <= j < N[1] and 0 <= k < N[2]):
                    continue
                v = float(geo.phi[i, j, k]*geo.K_x[i, j, k])
                if v > best:
                    best, arg = v, (i, j, k)
    # 0.2 m off the centre in x: a point within the well radius (0.1 m) of a cell centre would spread the delta over every
    # layer within 1 m of it (wellcase.py:141-155), dead cells included; off-centre, the nearest-centre rule picks this cell
    return [(arg[0] + 0.5)*D[0] + 0.2, (arg[1] + 0.5)*D[1], (arg[2] + 0.5)*D[2]]


def make_model(name, engine_factory=None, Nxyz=None, **over):
    params, geo, case, cls, kw = build_case(name, Nxyz)
    kw.update(over)
    return cls(geo, case, params, end=1e9, filename=None, verbosity=False, _engine_factory=engine_factory, **kw)


_BEAT = [0.0]


def heartbeat(what, model, every=30.0):
    """One progress line on stderr every `every` seconds (rank 0): long runs (config 5) must not look hung to the launcher."""
    now = time.perf_counter()
    if now - _BEAT[0] >= every and int(os.environ.get("RANK", "0")) == 0:
        _BEAT[0] = now
        print("[bench] %s: time step %d, dt %.4g d, Newton its %d, FGMRES its %d, failed solves %d"
              % (what, model.i_step, float(model.dt)/86400.0, model.total_nits, model.total_lits, model.failed_solves),
              file=sys.stderr, flush=True)


def spin_up(model, cap):
    """The reference's own dt ramp from maxdt*dt_init_fact up to maxdt (untimed initialisation of the state)."""
    maxdt_s = model.maxdt*86400.0
    n = 0
    t0 = time.perf_counter()
    while float(model.dt) < maxdt_s*(1.0 - 1e-12) and n < cap:
        model.step()
        n += 1
        heartbeat("spin-up", model)
    return n, time.perf_counter() - t0


def cpu_baseline(name, Nxyz, u, u_old, dt, budget_s=90.0, over=None):
    """oracle/cport (C++/OpenMP restatement of the same algorithm; test infrastructure, never the product) on the
    GPU box's host cores: the Newton solve of the FIRST timed time step -- same state, same old state, same dt,
    same tolerances.  Every leg runs that WHOLE solve (the reference's unit of work: one ``solver.solve()``,
    thermalmodel.py:165); `budget_s` is only a safety net that stops a leg between Newton iterations (reported as
    ``complete: false``).  Legs: the CPU share of this box (TP_CPU_THREADS, default min(CPUs, 16)) TWICE -- the run-to-run
    spread is part of the answer --, every CPU the process may run on when that is more (``all_visible``), and one thread."""
    from oracle import cport
    m = make_model(name, engine_factory=cport.CPortEngine, Nxyz=Nxyz, **(over or {}))
    eng = m.engine
    share = cport.default_threads()
    visible = cport._NCPU
    quota = _cpu_quota()
    plan = [("share_a", share), ("share_b", share)]
    # an all-cores leg only where the process may really use more CPUs than `share`: on the one-GPU boxes of this pool 256
    # hardware threads are VISIBLE but the cgroup quota is 16 CPUs -- 256 pinned OpenMP threads on a 16-CPU quota did not
    # finish one Newton solve in 7 minutes (round 3), so there the share legs ARE the all-core legs
    usable = visible if quota is None else min(visible, int(quota))
    if usable > share:
        plan.append(("all_visible", usable))
    plan.append(("t1", 1))
    legs = {}
    for label, nthreads in plan:
        eng.set_threads(nthreads)
        eng.set_state(u)
        eng.set_old(u_old)
        eng.set_dt(dt)
        r = eng.newton_solve(budget_s=budget_s)
        legs[label] = dict(threads=nthreads, seconds=r["seconds"], newton_its=r["nits_done"], fgmres_its=r["lits"],
                           newton_per_s=r["nits_done"]/r["seconds"], fgmres_per_s=r["lits"]/r["seconds"],
                           complete=bool(r["complete"]))
        print("[bench] cpu leg %s: %d threads, %.1f s, %d Newton / %d FGMRES its%s" % (
            label, nthreads, r["seconds"], r["nits_done"], r["lits"], "" if r["complete"] else " (cut by the safety budget)"),
            file=sys.stderr, flush=True)
    pair = [legs["share_a"], legs["share_b"]]
    best = max(pair, key=lambda q: q["newton_per_s"])
    out = {"value": best["newton_per_s"], "unit": "Newton steps/s", "cores": best["threads"], "kind": "port",
           "fgmres_its_per_s": best["fgmres_per_s"],
           "spread": {"runs": [q["newton_per_s"] for q in pair],
                      "note": "the same %d-thread leg twice, back to back; value = the faster one" % share},
           "one_thread": {"value": legs["t1"]["newton_per_s"], "fgmres_its_per_s": legs["t1"]["fgmres_per_s"],
                          "complete": legs["t1"]["complete"]},
           "cpu_model": _cpu_model(), "cpus_visible": visible, "cpu_quota": quota,
           "legs": legs,
           "sample": "oracle/cport (C++/OpenMP restatement of the reference algorithm, f64) on the same case: the whole Newton "
                     "solve of the first timed time step (same state, dt %.4g d): %d threads %.1f s and %.1f s (%d Newton / %d "
                     "FGMRES its), 1 thread %.1f s%s"
                     % (dt/86400.0, share, pair[0]["seconds"], pair[1]["seconds"], best["newton_its"], best["fgmres_its"],
                        legs["t1"]["seconds"], "" if legs["t1"]["complete"] else " (cut by the safety budget)")}
    if "all_visible" in legs:
        out["all_visible"] = {"value": legs["all_visible"]["newton_per_s"], "cores": usable,
                              "note": "every CPU the process may use (visible CPUs capped by the cgroup quota)"}
    return out


def _cpu_quota():
    """CPUs granted by the cgroup (cpu.max = "quota period"), None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q)/float(per)
    except (OSError, ValueError):
        return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # (the driver's command; the whole default run takes about a minute)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=90.0, help="safety budget (s) per cpu_baseline leg; a leg normally runs its whole Newton solve")
    ap.add_argument("--long-steps", type=int, default=40,
                    help="extra time steps run (and timed separately) AFTER the K timed ones for the failure-inclusive "
                         "long-window rate (0: off); `value` is always the K-step figure")
    ap.add_argument("--spinup-cap", type=int, default=80, help="max time steps of the untimed dt ramp")
    ap.add_argument("--grid", type=int, nargs=3, default=None, help="override Nx Ny Nz (development only)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="engine option override for experiments (e.g. amg_full_levels=2, amg_nu=1); recorded in config.workload")
    ap.add_argument("--preset", default=None, help="solver_parameters preset instead of the configuration's own (e.g. "
                    "pc_cptramg_QI, pc_cprilu1_gmres, pc_cptr_a11): measured alternatives, never the headline")
    ap.add_argument("--save-state", default=None, help="write the state at the start of the timed region (.npz: u, dt)")
    args = ap.parse_args()

    import torch
    from thermalporous_amd import parallel
    from thermalporous_amd.engine import EngineError
    rank, world = parallel.world()
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    if world > 1:
        parallel.init("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")

    Nxyz = tuple(args.grid) if args.grid else None
    if args.config == "c5" and Nxyz is None and world < 4:
        raise SystemExit("--config c5 is the 240x880x340 box (71.8 M cells, ~35 GB of HBM per 9 M-cell slab): launch it on 8 "
                         "GPUs (4 at least); --config c5slab runs one of its eight slabs on one GPU")
    over = {"solver_parameters": args.preset} if args.preset else {}
    model = make_model(args.config, Nxyz=Nxyz, **over)
    eng = model.engine
    if args.opt:
        kw = {}
        for kv in args.opt:
            k, v = kv.split("=", 1)
            kw[k] = eval(v, {"__builtins__": {}}, {"True": True, "False": False, "None": None})
        eng.set_options(**kw)
    model.start()
    # ---- spin-up: the reference's dt ramp, untimed ------------------------------------------------------
    n_spin, t_spin = spin_up(model, args.spinup_cap)
    ramp = {"steps": n_spin, "newton_its": model.total_nits, "fgmres_its": model.total_lits,
            "failed_solves": model.failed_solves, "solve_seconds": float(sum(model.timings)),
            "newton_per_s": model.total_nits/max(sum(model.timings), 1e-300),
            "fgmres_per_s": model.total_lits/max(sum(model.timings), 1e-300),
            "dt_days": [model.dt_vec[0]/86400.0, model.dt_vec[-1]/86400.0] if model.dt_vec else None}
    for _ in range(args.warmup):
        model.step()
        heartbeat("warm-up", model)
    n0, l0, f0, s0 = model.total_nits, model.total_lits, model.failed_solves, len(model.dt_vec)
    # state at the start of the timed region, for the CPU leg (same state, same dt)
    want_cpu = (not args.no_cpu_baseline) and world == 1
    if args.save_state and rank == 0 and world == 1:      # (at the start of a time step u == u_old)
        np.savez(args.save_state, u=eng.get_state(), dt=float(model.dt))
    if want_cpu:
        u_start = eng.get_state().copy()
        uold_start = eng.get_old_state().copy()
        dt_start = float(model.dt)
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.step()
        heartbeat("timed region", model)
    torch.cuda.synchronize()
    parallel.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    nits, lits = model.total_nits - n0, model.total_lits - l0
    dts = np.array(model.dt_vec[s0:])/86400.0
    # ---- long window: the K timed steps plus `long_steps` more, failed solves included (one time step in ~15 fails at
    # dt = 0.1 d and throws 25 Newton iterations away: a 20-step window may or may not contain one) -------------------
    long_window = None
    if args.long_steps > 0:
        f1 = model.failed_solves
        t1 = time.perf_counter()
        for _ in range(args.long_steps):
            model.step()
            heartbeat("long window", model)
        torch.cuda.synchronize()
        parallel.barrier()
        el2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el2], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el2 = float(t.item())
        nl, ll = model.total_nits - n0, model.total_lits - l0
        long_window = {"steps": args.steps + args.long_steps, "seconds": el + el2, "newton_its": nl, "fgmres_its": ll,
                       "newton_per_s": nl/(el + el2), "fgmres_per_s": ll/(el + el2),
                       "failed_solves": model.failed_solves - f0,
                       "dt_days": [float(np.min(model.dt_vec[s0:])/86400.0), float(np.max(model.dt_vec[s0:])/86400.0)],
                       "note": "the K timed steps plus %d more, timed back to back; failed solves (their wall time, not their "
                               "iterations) included" % args.long_steps}
        del f1

    # ---- per-kernel HIP-event timings on the library's stream, on the final Jacobian of the timed region ------
    eng._ck(eng.lib.tp_jacobian(eng.ctx))
    eng.pc_setup()
    ncell_local = eng.n[0]*eng.n[1]*eng.n[2]
    key = (eng.b, 7 if eng.gn[2] > 1 else 5)
    km = {nm: eng.time_kernel(w, reps) for w, nm, reps in ((0, "spmv_ms", 50), (1, "ilu_solve_ms", 50), (2, "amg_vcycle_ms", 20),
                                                           (3, "assembly_ms", 20), (4, "pc_apply_ms", 20), (5, "pc_setup_ms", 20),
                                                           (6, "ilu_factor_ms", 20))}

    def line(bytes_per_cell, ms):
        gbs = bytes_per_cell*ncell_local/(ms*1e-3)/1e9
        return {"bytes_per_cell": bytes_per_cell, "avg_ms": ms, "achieved_GBs": gbs, "frac": gbs/HBM_PEAK_GBS}
    # one classical Gram-Schmidt step at k = 16 basis vectors: VecMDot reads 16 V + w, VecMAXPY reads 16 V + w and writes w
    try:
        km["gram_schmidt_k16_ms"] = eng.time_kernel(7, 30)
        gs = line((2*16 + 3)*eng.b*8, km["gram_schmidt_k16_ms"])
    except EngineError as e:          # only "no solve has used 17 basis vectors yet: nothing to time"; anything else is real
        if "Krylov basis smaller than 17" not in str(e):
            raise
        gs = None
    others = {"spmv_block": line(SPMV_BYTES_PER_CELL[key], km["spmv_ms"]),
              "assembly_residual_jacobian": line(ASM_BYTES_PER_CELL[key], km["assembly_ms"]),
              "ilu0_factor": line(FACTOR_BYTES_PER_CELL[key], km["ilu_factor_ms"])}
    if gs is not None:
        others["gram_schmidt_k16"] = gs
    # the reference's own rate definition (thermalmodel.py:395-403): iterations over the summed wall time of the
    # SUCCESSFUL solver.solve() calls of the timed steps (a failed solve's time is not in `timings` there either)
    tim = model.timings[:len(model.timings) - args.long_steps] if args.long_steps > 0 else model.timings
    t_ok = float(sum(tim[-args.steps:])) if len(tim) >= args.steps else None
    if rank != 0:
        return
    ilu = line(SPMV_BYTES_PER_CELL[key], km["ilu_solve_ms"])
    # one whole Krylov iteration (pc_apply + SpMV + Gram-Schmidt) as achieved algorithmic bytes/s: for the small 2-D
    # configurations, which are latency bound, this -- not a kernel's roofline fraction -- is the honest figure
    it_ms = 1e3*el/max(lits, 1)
    # HBM traffic per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950
    # correction applied; profiles/r02_pmc_traffic.json "how"): quoted only when the profile is of this launch shape
    traffic, traffic_source = None, None
    for prof_name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", prof_name)))
            if prof["cells"] == ncell_local and eng.b == 3:
                for kname, rec in prof["kernels"].items():
                    if "k_ilu_solve" in kname:
                        traffic = rec["traffic_bytes"]
                if traffic is not None:
                    traffic_source = "profiles/" + prof_name + " (rocprofv3 --pmc passes of an earlier run of this launch shape; NOT measured in this run)"
                    break
        except (OSError, KeyError, ValueError):
            pass
    # algorithmic bytes of ONE Krylov iteration (SURVEY.md 8d figures): block SpMV + ILU application (same bytes) + the
    # inter-stage residual on the primary columns + classical Gram-Schmidt at the average basis size + the V-cycles of
    # stage 1 (4.5 x 104 B x operator complexity per full cycle; a relaxation-only cycle = 2 sweeps of 104 B)
    pc = eng.opts["pc"]
    npri = 1 if pc == "cpr" else 2
    k_avg = 0.5*(lits/max(nits, 1) + 1.0)
    it_bytes = {"spmv": SPMV_BYTES_PER_CELL[key], "gram_schmidt": (2.0*k_avg + 3.0)*eng.b*8}
    if pc != "fieldsplit_cd":
        it_bytes["ilu_sweeps"] = SPMV_BYTES_PER_CELL[key]
        it_bytes["stage_residual"] = SPMV_BYTES_PER_CELL[key]*npri/eng.b
    vc = 0.0
    for which, count in ((0, 2 if pc in ("cptr", "fieldsplit_cd") else 1), (1, 1 if pc in ("cptr", "fieldsplit_cd") else 0)):
        if count:
            try:
                _, oc = eng.amg_info(which)
                tr, _ = eng.amg_trunc(which)
                vc += count*(2*104.0 if tr == 0 else 4.5*104.0*oc)
            except Exception:
                pass
    it_bytes["vcycles"] = vc
    it_total = float(sum(it_bytes.values()))
    whole = {"bytes_per_cell": it_total, "breakdown": it_bytes, "ms": it_ms,
             "achieved_GBs": it_total*ncell_local/(it_ms*1e-3)/1e9, "frac": it_total*ncell_local/(it_ms*1e-3)/1e9/HBM_PEAK_GBS,
             "note": "algorithmic bytes of one FGMRES iteration / wall time per iteration of the timed region (everything "
                     "included: Newton overheads, host syncs, failed solves); the honest figure for the latency-bound 2-D configs"}
    desc = CONFIGS[args.config][0]
    out = {
        "metric": "Newton steps/s, SPE10 60x220x85 two-phase (FGMRES its/s in config)" if args.config == "c4"
                  else "Newton steps/s (FGMRES its/s in config), " + args.config,
        "value": nits/el,
        "unit": "Newton steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3*el/args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s; grid %dx%dx%d (synthetic default_rng(10) field); FGMRES rtol %g; timed at dt = maxdt after an "
                        "untimed spin-up along the reference's dt ramp%s" % (desc, model.geo.Nx, model.geo.Ny, model.geo.Nz,
                                                                           eng.opts["ksp_rtol"],
                                                                           ("; PRESET %s instead of the configuration's own" % args.preset if args.preset else "") +
                                                                           ("; OPTIONS %s" % ",".join(args.opt) if args.opt else "")),
            "fgmres_its_per_s": lits/el,
            "newton_its": nits, "fgmres_its": lits, "failed_solves": model.failed_solves - f0,
            "dt_days": [float(dts.min()), float(dts.max())],
            "ms_per_fgmres_it": it_ms,
            "reference_style_rates": None if not t_ok else {"newton_per_s": nits/t_ok, "fgmres_per_s": lits/t_ok,
                                                           "note": "sum(nits)/sum(timings) of successful solves only, "
                                                                   "as thermalmodel.py:395-403 prints it"},
            "ramp": ramp,
            "long_window": long_window,
            "slabs": "1-D along the slab axis" if world > 1 else "none",
            "kernels_ms": km,
        },
        "roofline": {"bound": "hbm", "achieved": ilu["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ilu["frac"], "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": ("k_ilu1_solve<%d>" if eng.opts.get("ilu_levels") else
                                "k_ilu_solve_mw<%d> x tile-diagonal launches (ilu_whole)" if eng.opts.get("ilu_whole") else
                                "k_ilu_solve<%d>" if os.environ.get("TP_ILU_MW") == "0" else "k_ilu_solve_mw<%d>") % eng.b,
                     "bytes_per_cell": ilu["bytes_per_cell"],
                     "cells_per_launch": ncell_local, "avg_ms": ilu["avg_ms"], "other_kernels": others,
                     "whole_iteration": whole},
    }
    if want_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(args.config, Nxyz, u_start, uold_start, dt_start, args.cpu_budget, over)
        except Exception as e:          # the CPU leg must never take the GPU measurement down with it
            out["cpu_baseline"] = {"value": None, "unit": "Newton steps/s", "cores": 0, "kind": "port",
                                   "sample": "cpu leg failed: %r" % (e,)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
