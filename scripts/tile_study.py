"""Krylov iterations of stage 2 = bjacobi tiles vs ONE block per GPU (the reference's 1-rank operator,
sub_1_pc_bjacobi_blocks 1 / singlephase.py:348-349), on full-size C4 in the 0.1-day regime.

Runs on the CPU with oracle/cport (the GPU engine cannot sweep a whole-slab tile with one wavefront) from the state
bench.py saved at the start of its timed region:   python bench.py --no-cpu-baseline --save-state gpurun_out/c4_state.npz
Usage: python scripts/tile_study.py gpurun_out/c4_state.npz [dt=SECONDS] [tile ...]      tile = t0,t1,t2
(dt: override the saved time step -- the saved dt = maxdt step may be one the time loop had to halve)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                     # noqa: E402
from oracle import cport                         # noqa: E402

st = np.load(sys.argv[1])
u, dt = st["u"], float(st["dt"])
args = sys.argv[2:]
if args and args[0].startswith("dt="):
    dt = float(args.pop(0)[3:])
tiles = [tuple(int(v) for v in a.split(",")) for a in args] or [(1 << 30, 8, 8), (1 << 30, 16, 16), (1 << 30, 1 << 30, 1 << 30)]
out = []
for tile in tiles:
    # same options as the bench run, only the bjacobi tile differs
    m = bench.make_model("c4", engine_factory=lambda spec, opts: cport.CPortEngine(spec, dict(opts, ilu_tile=tile)))
    e = m.engine
    e.set_state(u)
    e.set_old(u)
    e.set_dt(dt)
    t0 = time.time()
    r = e.newton_solve()
    rec = dict(tile=[min(t, 9999) for t in tile], ntiles=e.ntiles(), newton_its=r["nits"], fgmres_its=r["lits"],
               reason=r["reason"], seconds=round(time.time() - t0, 1), dt_days=dt/86400.0)
    print(json.dumps(rec), flush=True)
    out.append(rec)
