import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
m = bench.make_model(sys.argv[1] if len(sys.argv) > 1 else "c4")
m.start()
e = m.engine
e.set_old(None); e.set_dt(100.0)
e._ck(e.lib.tp_jacobian(e.ctx)); e.pc_setup()
print("ilu_solve_ms %.4f" % e.time_kernel(1, 50), flush=True)
