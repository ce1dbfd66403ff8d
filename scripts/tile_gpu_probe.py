"""Development probe: bjacobi tile shapes on the GPU -- ILU sweep time, pc_apply, Krylov counts and wall time of the Newton
solve of the first timed step of a bench configuration (state after the untimed spin-up)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from thermalporous_amd.engine import HipEngine
cfg = sys.argv[1]
tiles = [("L1",) if a == "L1" else tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]        # L1: block-ILU(1), default tile
m = bench.make_model(cfg)
m.start()
bench.spin_up(m, 200)
u = m.engine.get_state().copy()
dt = float(m.dt)
m.engine.close()
for tile in [None] + tiles:
    over = {} if not tile else dict(ilu_levels=1) if tile == ("L1",) else dict(ilu_tile=tile)
    f = (lambda spec, opts: HipEngine(spec, dict(opts, **over))) if tile else None
    mm = bench.make_model(cfg, engine_factory=f)
    e = mm.engine
    for rep in range(2):
        e.set_state(u); e.set_old(u); e.set_dt(dt)
        t0 = time.perf_counter()
        r = e.newton_solve()
        wall = time.perf_counter() - t0
    e._ck(e.lib.tp_jacobian(e.ctx)); e.pc_setup()
    print(json.dumps(dict(levels=e.opts.get("ilu_levels", 0), tile=[min(t, 9999) for t in e.opts["ilu_tile"]], nits=r["nits"], lits=r["lits"], reason=r["reason"], solve_ms=round(wall*1e3, 1),
                          ilu_solve_ms=round(e.time_kernel(1, 50), 4), pc_apply_ms=round(e.time_kernel(4, 20), 4), ilu_factor_ms=round(e.time_kernel(6, 10), 4))), flush=True)
    e.close()
