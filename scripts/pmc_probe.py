"""Runs each hot kernel of the C4 workload a few times (for rocprofv3 --pmc passes / kernel traces)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
m = bench.make_model("c4")
m.start()
for _ in range(14):          # up the dt ramp: a realistic state, and a Krylov basis of > 17 vectors for the Gram-Schmidt probe (7)
    m.step()
e = m.engine
e._ck(e.lib.tp_jacobian(e.ctx))
e.pc_setup()
for w in (0, 1, 3, 4, 6, 7):      # spmv, ilu solve, assembly, pc_apply (eager under TP_GRAPH=0), ilu factor
    e.time_kernel(w, 5)
print("done")
