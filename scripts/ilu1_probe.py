"""Development probe: block-ILU(0) vs block-ILU(1) second stage on a bench configuration (Krylov counts, kernel times)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
base = bench.make_model(cfg)
sp = dict(base.solver_parameters)
base.engine.close()
for lev in (0, 1):
    m = bench.make_model(cfg, solver_parameters={**sp, "sub_1_sub_pc_factor_levels": lev})
    m.start()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        m.step()
    dt = time.perf_counter() - t0
    e = m.engine
    print(cfg, "levels", lev, "steps %d wall %.2f s" % (nsteps, dt), "newton", getattr(m, "total_nits", None), "lits", getattr(m, "total_lits", None),
          "ilu_solve %.4f ms ilu_factor %.4f ms pc_apply %.4f ms" % (e.time_kernel(1, 50), e.time_kernel(6, 10), e.time_kernel(4, 20)), flush=True)
    e.close()
