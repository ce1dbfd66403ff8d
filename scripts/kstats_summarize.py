"""profiles/rNN_bench_kernel_stats_summary.txt from a rocprofv3 --kernel-trace --stats kernel_stats.csv (calls, average, share)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[2] if len(sys.argv) > 2 else "")
print("%-84s %8s %10s %7s" % ("kernel", "calls", "avg_us", "share"))
grp = {}
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    nm = r["Name"]
    share = float(r["TotalDurationNs"])/tot
    print("%-84s %8d %10.1f %6.2f%%" % (nm[:84], int(r["Calls"]), float(r["AverageNs"])/1e3, 100*share))
    key = ("AMG (k_amg_*) + scalar SpMV" if ("k_amg" in nm or "k_spmv_scalar" in nm) else "ILU solve" if "k_ilu_solve" in nm
           else "Gram-Schmidt (k_multi_dot + k_multi_axpy_norm)" if ("k_multi_dot" in nm or "k_multi_axpy_norm" in nm)
           else "block SpMV (both variants)" if "k_spmv_block" in nm else "ILU factorisation (gather + factor)" if "k_ilu" in nm
           else "assembly" if ("k_assemble" in nm or "k_sources" in nm) else "other")
    grp[key] = grp.get(key, 0.0) + share
print("# shares: " + ", ".join("%s %.1f %%" % (k, 100*v) for k, v in sorted(grp.items(), key=lambda kv: -kv[1])))
print("# total kernel time %.3f s" % (tot/1e9))
