"""Development probe: ILU tile width for the 2-D configurations (C3 60x220 two-phase, pc_cptr)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import cases
from thermalporous_amd.twophase import TwoPhase
for t1 in (64, 32, 16, 8):
    spec, u0, p, g, c = cases.c3_spe10_2d(60, 220, 2)
    m = TwoPhase(g, c, p, end=4.0, maxdt=1.0, solver_parameters="pc_cptr", filename=None, verbosity=False)
    m.engine.set_options(ilu_tile=(1 << 30, t1, 1))
    t = time.time()
    m.solve()
    e = m.engine
    print("t1 %3d  steps %3d newton %4d krylov %5d failed %d  %.3fs   ilu %.3f pc_apply %.3f ms" % (
        t1, len(m.dt_vec), m.total_nits, m.total_lits, m.failed_solves, time.time() - t, e.time_kernel(1, 20), e.time_kernel(4, 20)), flush=True)
