"""Development probe: the N-slab algorithm (in-process slab group, ONE GPU) on a large grid.
usage: slab_probe.py Nx Ny Nz nranks [nsteps] [opts-dict]"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench
from thermalporous_amd import engine as E

if len(sys.argv) in (2, 3, 4) or (len(sys.argv) > 1 and sys.argv[1] in ("-h", "--help")):
    raise SystemExit(__doc__)
if len(sys.argv) == 1:                       # no arguments: C4's own geometry on 4 in-process slabs
    sys.argv += ["60", "220", "85", "4"]
Nxyz = tuple(int(v) for v in sys.argv[1:4])
nranks = int(sys.argv[4])
nsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
extra = eval(sys.argv[6]) if len(sys.argv) > 6 else {}
m = bench.make_model("c4", Nxyz=Nxyz)
spec, opts = m.spec, dict(m.engine_opts, **extra)
m.start()
m.u.flush() if hasattr(m.u, "flush") else None
u0 = m.engine.get_state().copy()
m.engine.close()
dt0 = float(os.environ.get('DT0', 86400.0*m.maxdt*m.dt_init_fact))
dts = [dt0*2.0**k for k in range(nsteps)]
print("grid", Nxyz, "cells", np.prod(Nxyz), "ranks", nranks, "dts", dts, flush=True)

lib = E.load_library()
group = C.c_void_p()
if nranks > 1:
    assert lib.tp_local_group_create(nranks, C.byref(group)) == 0
out = [None]*nranks


def worker(rank):
    h = E.HipEngine(spec, opts, rank=rank, nranks=nranks, local_group=group if nranks > 1 else None)
    h.set_state(u0)
    infos = []
    t0 = time.time()
    for dt in dts:
        h.set_old(None)
        h.set_dt(dt)
        infos.append(h.newton_solve())
    el = time.time() - t0
    if rank == 0:
        print("newton", [(i["nits"], i["lits"], i["reason"]) for i in infos], "%.2fs" % el, flush=True)
    lay = h.amg_layout(0)
    kt = {nm: h.time_kernel(w, 5) for w, nm in ((0, "spmv"), (1, "ilu"), (2, "vcycle"), (4, "pc_apply"))}
    out[rank] = (infos, el, lay, kt, h.get_state())
    h.close()


ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
for t in ts:
    t.start()
for t in ts:
    t.join()
infos, el, lay, kt, _ = out[0]
print("dist_levels", lay[0], "sched", lay[1])
print("newton", [(i["nits"], i["lits"], i["reason"]) for i in infos], "%.2fs" % el)
print("kernels_ms", {k: round(v, 3) for k, v in kt.items()})
state = np.concatenate([o[4] for o in out], axis=1)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "slab_probe_%d.npy" % nranks), state[:, ::8, ::8, ::8])
