"""Per-(kernel, grid size) duration table from a rocprofv3 --kernel-trace CSV directory (development aid)."""
import csv, glob, statistics, sys
rows = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tp::", "")
        key = (name, int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])*int(r.get("Grid_Size_Y", 1))*int(r.get("Grid_Size_Z", 1)))
        rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (name, grid), v in sorted(rows.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    if flt in name:
        print("%-46s grid %9d  n %6d  median %8.2f us  min %8.2f" % (name[:46], grid, len(v), statistics.median(v), min(v)))
