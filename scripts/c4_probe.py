"""Development probe: C4 (60x220x85 two-phase) on one GPU -- kernel timings and a few time steps."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import cases
from thermalporous_amd.engine import HipEngine

Nx, Ny, Nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (60, 220, 85)
spec, u0, p, g, c = cases.c4_spe10_3d(Nx, Ny, Nz)
opts = dict(pc="cptr", ksp_rtol=1e-8, snes_max_it=25)
t = time.time()
h = HipEngine(spec, opts)
print("engine setup %.2fs" % (time.time() - t), "n", spec["n"], flush=True)
h.set_state(u0)
ncell = np.prod(spec["n"])
dts = [1e-3, 2e-3, 4e-3, 8e-3, 1.6e-2, 3.2e-2]
tot_n = tot_l = 0; tot_t = 0.0
for i, d in enumerate(dts):
    h.set_old(None); h.set_dt(d*86400)
    t = time.time(); r = h.newton_solve(); el = time.time() - t
    print("step %d dt=%g nits=%d lits=%d reason=%d fnorm=%.2e t=%.3fs  (%.1f ms/lit)" % (i, d, r["nits"], r["lits"], r["reason"], r["fnorm"], el, 1e3*el/max(1, r["lits"])), flush=True)
    if i > 0 and r["reason"] > 0:
        tot_n += r["nits"]; tot_l += r["lits"]; tot_t += el
print("Newton steps/s %.2f  FGMRES its/s %.1f" % (tot_n/tot_t, tot_l/tot_t))
h.jacobian(); h.pc_setup()
names = ["spmv", "ilu_solve", "amg_vcycle", "assembly", "pc_apply"]
bytes_per_cell = [584, 584, None, 616, None]
for w, nm in enumerate(names):
    ms = h.time_kernel(w, 20)
    s = "%-12s %.3f ms" % (nm, ms)
    if bytes_per_cell[w]:
        s += "  %.0f GB/s algorithmic" % (bytes_per_cell[w]*ncell/ms/1e6)
    print(s, flush=True)
print("amg levels, opc", h.amg_info(0))
