"""Development probe: the first time steps of bench.py --config c5slab (dt, Newton / Krylov counts, failures)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
grid = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else None
m = bench.make_model("c5slab", Nxyz=grid)
m.start()
for i in range(int(os.environ.get("NSTEPS", "14"))):
    f0 = m.failed_solves
    dt = float(m.dt)
    t = time.time()
    n, l = m.step()
    info = m.engine.last
    print("step %2d dt %.4g s -> used %.4g s  nits %d lits %d failed %d  fnorm0 %.3g fnorm %.3g reason %s  %.2fs" % (
        i, dt, m.dt_vec[-1], n, l, m.failed_solves - f0, info.get("fnorm0", 0), info.get("fnorm", 0), info.get("reason"), time.time() - t), flush=True)
