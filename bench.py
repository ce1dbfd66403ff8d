#!/usr/bin/env python3
"""bench.py -- Newton steps/s (+ FGMRES its/s) of the hot path on BASELINE config 4:
two-phase 3-D SPE10-like 60x220x85 box, wells + heaters, pc_cptr (CPTR: fieldsplit-Schur stage 1 with
AMG V-cycles on App and S~, block-Jacobi block-ILU(0) stage 2) inside FGMRES inside Newton.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one time step of the reference's time loop = one ``solver.solve()`` (one Newton solve with
its linear solves) + the loop's dt policy (thermalporous_amd/thermalmodel.py:step).  All inputs are
synthetic (the SPE10 .dat files are not shipped with the reference) and resident in HBM before the
timed region.  N > 1: the fixed 60x220x85 problem is cut into N slabs along y ("strong" scaling).
Rank 0 prints ONE JSON line with the metric, a roofline object for the block SpMV kernel and the CPU
baseline (the numpy oracle, timed on a bounded sample; never the thing measured).
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


def build_case(name, Nxyz=None, refine=1):
    """BASELINE configs on synthetic data."""
    from thermalporous_amd.physicalparameters import PhysicalParameters
    from thermalporous_amd.SPE10model3D import SPE10Model3D
    from thermalporous_amd.wellheatercase import WellHeaterCase
    params = PhysicalParameters()
    params.rate = 2e-4          # tests_twophase/test_60x120_wells_default.py:8-9 of the reference
    params.S_o = 0.9
    params.T_inj = 373.15
    if name == "c4":
        Nx, Ny, Nz = Nxyz or (60, 220, 85)
    else:
        raise ValueError(name)
    geo = SPE10Model3D(Nx, Ny, Nz, params, refine=refine)    # refine r: the 60x220x85 field upsampled r-fold (BASELINE config 5: r=4)
    L, Ly, Lz = geo.Length, geo.Length_y, geo.Length_z
    # SPE10 well (x,y) positions (wellcase.py:30-36), producer low / injector high in the column
    prod = [[140.0/365.76*L, 210.0/670.56*Ly, 0.2*Lz]]
    inj = [[265.0/365.76*L, 260.0/670.56*Ly, 0.8*Lz]]
    case = WellHeaterCase(params, geo, prod_points=prod, inj_points=inj)
    return params, geo, case


def make_model(name, engine_factory=None, Nxyz=None, maxdt=0.1, refine=1):
    from thermalporous_amd.twophase import TwoPhase
    params, geo, case = build_case(name, Nxyz, refine)
    return TwoPhase(geo, case, params, end=1e9, maxdt=maxdt, small_dt_start=True, solver_parameters="pc_cptr",
                    filename=None, verbosity=False, _engine_factory=engine_factory)


def cpu_baseline(steps=1):
    """The numpy oracle (a "port" of the reference algorithm) on a bounded sample: the first time steps of
    the same case on a 30x110x43 sub-box (1/8 of the cells), one core.  Newton steps/s on the full box is
    estimated as the sample rate divided by 8 (cost is linear in cells); both numbers are reported."""
    from oracle.engine import OracleEngine
    frac = 8.0
    m = make_model("c4", engine_factory=OracleEngine, Nxyz=(30, 110, 43))
    m.start()
    t0 = time.perf_counter()
    nits = lits = 0
    for _ in range(steps):
        n, l = m.step()
        nits += n
        lits += l
    el = time.perf_counter() - t0
    return {"value": nits/el/frac, "unit": "Newton steps/s", "cores": 1, "kind": "port",
            "sample": "numpy oracle, first %d time step(s) of the same case on a 30x110x43 sub-box (1/8 of the cells): "
                      "%.3f Newton steps/s, %.2f FGMRES its/s measured in %.1f s; value = measured/8 (linear-in-cells "
                      "estimate for 60x220x85)" % (steps, nits/el, lits/el, el),
            "fgmres_its_per_s": lits/el/frac}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grid", type=int, nargs=3, default=None, help="override Nx Ny Nz (development only)")
    args = ap.parse_args()

    import torch
    from thermalporous_amd import parallel
    rank, world = parallel.world()
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    if world > 1:
        parallel.init("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")

    model = make_model(args.config, Nxyz=tuple(args.grid) if args.grid else None)
    eng = model.engine
    model.start()
    for _ in range(args.warmup):
        model.step()
    n0, l0, f0 = model.total_nits, model.total_lits, model.failed_solves
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.step()
    torch.cuda.synchronize()
    parallel.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    nits, lits = model.total_nits - n0, model.total_lits - l0

    # roofline of the dominant streaming kernel of one Krylov iteration: the 7-point 3x3-block SpMV
    eng._ck(eng.lib.tp_jacobian(eng.ctx))
    eng.pc_setup()
    ncell_local = eng.n[0]*eng.n[1]*eng.n[2]
    ms = eng.time_kernel(0, 50)
    bpc = SPMV_BYTES_PER_CELL[(eng.b, 7 if eng.gn[2] > 1 else 5)]
    achieved = bpc*ncell_local/(ms*1e-3)/1e9
    extra = {nm: eng.time_kernel(w, 20) for w, nm in ((1, "ilu_solve_ms"), (2, "amg_vcycle_ms"), (3, "assembly_ms"),
                                                       (4, "pc_apply_ms"), (5, "pc_setup_ms"), (6, "ilu_factor_ms"))}
    # the other streaming kernels against the same roofline (algorithmic bytes per cell: SURVEY.md 8d)
    others = {}
    if eng.b == 3 and eng.gn[2] > 1:
        for nm, key, bytes_per_cell in (("ilu0_solve", "ilu_solve_ms", 584), ("assembly_residual_jacobian", "assembly_ms", 616),
                                        ("ilu0_factor(gather+factor)", "ilu_factor_ms", 1040)):
            gbs = bytes_per_cell*ncell_local/(extra[key]*1e-3)/1e9
            others[nm] = {"bytes_per_cell": bytes_per_cell, "avg_ms": extra[key], "achieved_GBs": gbs, "frac": gbs/HBM_PEAK_GBS}
    if rank != 0:
        return
    # HBM traffic per launch of the same kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs, gfx950 correction applied; see profiles/r01_pmc_traffic.json "how"): only quoted when the
    # committed profile is of this very launch shape
    traffic = None
    try:
        prof = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")))
        if prof["cells"] == ncell_local and eng.b == 3:
            traffic = prof["kernels"]["tp::k_spmv_block<3, 7, 3, 0>"]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "Newton steps/s, SPE10 60x220x85 two-phase (FGMRES its/s in config)",
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
            "workload": "BASELINE config 4: two-phase 3D SPE10-like %dx%dx%d (synthetic default_rng(10) field), wells+heaters, "
                        "pc_cptr, FGMRES rtol 1e-8, dt ramp from maxdt*2^-10 (maxdt 0.1 d)" % (model.geo.Nx, model.geo.Ny, model.geo.Nz),
            "fgmres_its_per_s": lits/el,
            "newton_its": nits, "fgmres_its": lits, "failed_solves": model.failed_solves - f0,
            "dt_days": [model.dt_vec[args.warmup]/86400.0, model.dt_vec[-1]/86400.0],
            "slabs": "1-D along y" if world > 1 else "none",
            "kernels_ms": dict(spmv_ms=ms, **extra),
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved/HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "k_spmv_block<3,7,3,0>", "bytes_per_cell": bpc, "cells_per_launch": ncell_local,
                     "avg_ms": ms, "other_kernels": others},
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
