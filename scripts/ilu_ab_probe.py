"""Development probe: ILU factor / solve timings of two builds of the library on the same box (A/B)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from thermalporous_amd import engine
import bench
cfg = sys.argv[1]
for tag, path in (("new", None), ("old", os.path.join(os.path.dirname(__file__), "..", "gpurun_probe_old.so"))):
    if path and not os.path.exists(path):
        continue
    engine._LIB = engine.load_library(path) if path else None
    m = bench.make_model(cfg)
    m.start()
    for _ in range(3):
        m.step()
    e = m.engine
    e._ck(e.lib.tp_jacobian(e.ctx)); e.pc_setup()
    print(cfg, tag, "ilu_solve %.4f ilu_factor %.4f pc_apply %.4f" % (e.time_kernel(1, 50), e.time_kernel(6, 30), e.time_kernel(4, 20)), flush=True)
    e.close()
