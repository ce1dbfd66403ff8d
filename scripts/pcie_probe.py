"""Development probe: the bench loop with a host round trip of the state every time step (PCIe-inclusive rate)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
for roundtrip in (False, True):
    m = bench.make_model("c4")
    m.start()
    for _ in range(2):
        m.step()
    n0 = m.total_nits
    t = time.perf_counter()
    for _ in range(8):
        m.step()
        if roundtrip:
            u = m.engine.get_state()
            m.engine.set_state(u)
    el = time.perf_counter() - t
    print("host round trip each step: %-5s  %.2f Newton steps/s  (%.1f ms/step)" % (roundtrip, (m.total_nits - n0)/el, 1e3*el/8), flush=True)
    m.engine.close()
