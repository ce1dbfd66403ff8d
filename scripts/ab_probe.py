"""Development probe: per-kernel HIP-event timings on C4 after 14 ramp steps (one JSON line; used for A/B runs under
different TP_* environment variables).  usage: ab_probe.py [label] [opts-dict]"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
label = sys.argv[1] if len(sys.argv) > 1 else "default"
m = bench.make_model("c4")
if len(sys.argv) > 2:
    m.engine.set_options(**eval(sys.argv[2]))
m.start()
for _ in range(int(os.environ.get("NSTEPS", "14"))):
    m.step()
e = m.engine
e._ck(e.lib.tp_jacobian(e.ctx))
e.pc_setup()
names = ["spmv", "ilu_solve", "amg_vcycle", "assembly", "pc_apply", "pc_setup", "ilu_factor", "gs_k16"]
out = {"label": label, "its": [m.total_nits, m.total_lits]}
for w, nm in enumerate(names):
    out[nm] = round(e.time_kernel(w, 30), 4)
print(json.dumps(out), flush=True)
