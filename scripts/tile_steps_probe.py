"""Development probe: bjacobi tile shapes through the TIME LOOP of a bench configuration (Krylov counts and wall time of
the first N time steps after the spin-up).  Usage: tile_steps_probe.py CONFIG NSTEPS [t0,t1,t2 ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from thermalporous_amd.engine import HipEngine
cfg, nsteps = sys.argv[1], int(sys.argv[2])
tiles = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]
for tile in [None] + tiles:
    f = (lambda spec, opts: HipEngine(spec, dict(opts, ilu_tile=tile))) if tile else None
    m = bench.make_model(cfg, engine_factory=f)
    m.start()
    bench.spin_up(m, 200)
    for _ in range(2):
        m.step()
    n0, l0, f0 = m.total_nits, m.total_lits, m.failed_solves
    t0 = time.perf_counter()
    for _ in range(nsteps):
        m.step()
    el = time.perf_counter() - t0
    e = m.engine
    print(json.dumps(dict(tile=[min(t, 9999) for t in e.opts["ilu_tile"]], nits=m.total_nits - n0, lits=m.total_lits - l0, failed=m.failed_solves - f0,
                          seconds=round(el, 4), newton_per_s=round((m.total_nits - n0)/el, 2), ms_per_it=round(1e3*el/max(1, m.total_lits - l0), 4))), flush=True)
    e.close()
