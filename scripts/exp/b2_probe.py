import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import cases
from thermalporous_amd.engine import HipEngine
for nph in (1,2):
    spec,u0,*_ = cases.c4_spe10_3d(60,220,85,nphase=nph)
    h = HipEngine(spec, dict(pc="cpr"))
    h.set_state(u0); h.set_old(None); h.set_dt(10.0)
    h.jacobian(); h.pc_setup()
    print("nphase", nph, "spmv %.3f ilu_solve %.3f ilu_factor %.3f ms" % (h.time_kernel(0,20), h.time_kernel(1,20), h.time_kernel(6,10)), flush=True)
    h.close()
