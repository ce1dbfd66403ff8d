"""Development probe: kernel timings on C4 after a few ramp steps."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
tile = eval(sys.argv[1]) if len(sys.argv) > 1 else None
m = bench.make_model("c4")
if tile:
    m.engine.set_options(ilu_tile=tile)
if len(sys.argv) > 2:
    m.engine.set_options(**eval(sys.argv[2]))
m.start()
t = time.time()
nsteps = int(os.environ.get("NSTEPS", "4"))
tot = [0, 0]
for i in range(nsteps):
    r = m.step()
    tot[0] += r[0]; tot[1] += r[1]
print("its", tot, flush=True)
print("steps %.3fs" % (time.time() - t), "failed", m.failed_solves)
e = m.engine
e.lib.tp_jacobian(e.ctx); e.pc_setup()
if os.environ.get("TP_DEBUG"):      # address map of the native libraries (to decode a crash stack)
    for ln in open("/proc/self/maps"):
        if " r-xp " in ln and any(k in ln for k in ("amdhip", "hsa-runtime", "rocprof", "thermalporous", "libc.so", "rocr", "roctx")):
            print("[maps]", ln.strip(), flush=True)
SYNC_EACH = os.environ.get("KT_ONLY")
for w, nm in enumerate(["spmv", "ilu_solve", "amg_vcycle", "assembly", "pc_apply", "pc_setup", "ilu_factor"]):
    if SYNC_EACH and nm not in SYNC_EACH.split(","):
        continue
    print("%-12s %.3f ms" % (nm, e.time_kernel(w, 20)), flush=True)
