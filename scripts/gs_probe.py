"""Development probe: one classical Gram-Schmidt step against 16 basis vectors on C4 (tp_time_kernel which=7)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
m = bench.make_model("c4")
m.start()
for _ in range(14):            # far enough up the ramp for solves with > 17 Krylov iterations
    m.step()
e = m.engine
e._ck(e.lib.tp_jacobian(e.ctx)); e.pc_setup()
ms = e.time_kernel(7, 50)
nbytes = (2*16 + 3)*e.b*8*e.n[0]*e.n[1]*e.n[2]
print("TP_MD_CHUNK=%s  gs16 %.4f ms  %.0f GB/s" % (os.environ.get("TP_MD_CHUNK", "8"), ms, nbytes/ms/1e6))
