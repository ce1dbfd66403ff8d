"""Development probe: Krylov iterations of the 2-D configurations for several AMG cycle shapes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import cases
from thermalporous_amd.twophase import TwoPhase
from thermalporous_amd.singlephase import SinglePhase

for nphase, preset in ((2, "pc_cptr"), (1, "pc_cpr"), (1, "pc_fieldsplit_cd")):
    for extra in ({}, {"amg_coarse_post": 2}, {"amg_full_levels": 99}):
        spec, u0, p, g, c = cases.c3_spe10_2d(60, 220, nphase)
        M = TwoPhase if nphase == 2 else SinglePhase
        m = M(g, c, p, end=2.0, maxdt=1.0, solver_parameters=preset, filename=None, verbosity=False)
        m.engine.set_options(**extra)
        t = time.time()
        m.solve()
        print("%d-phase %-18s %-26s steps %3d newton %4d krylov %5d failed %d  %.2fs" % (
            nphase, preset, extra, len(m.dt_vec), m.total_nits, m.total_lits, m.failed_solves, time.time() - t), flush=True)
