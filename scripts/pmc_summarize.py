"""Builds profiles/r02_pmc_traffic.json from two rocprofv3 PMC passes over scripts/pmc_probe.py:

  cd /tmp && export TMPDIR=/tmp && cd $REPO
  TP_GRAPH=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 scripts/pmc_probe.py
  TP_GRAPH=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 scripts/pmc_probe.py
  python3 scripts/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic.json

(separate passes, no other trace domain; TP_GRAPH=0 because --pmc crashes on hipGraph replay).  Counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so
traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch; medians over the sampled launches."""
import csv
import glob
import json
import statistics
import sys

CELLS = 60*220*85
KEEP = ("k_assemble", "k_spmv_block", "k_ilu_solve", "k_ilu_factor", "k_ilu_gather", "k_spmv_scalar", "k_multi_dot", "k_multi_axpy_norm")


def collect(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(k in name for k in KEEP):
                out.setdefault((name, int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    return out


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
kernels = {}
for key in sorted(fetch):
    if key not in write or key[1] < 10000:
        continue
    f, w = statistics.median(fetch[key]), statistics.median(write[key])
    t = (2*f + w)*1024
    kernels["%s [grid %d]" % key if sum(k[0] == key[0] for k in fetch) > 1 else key[0]] = {
        "fetch_size_kib": f, "write_size_kib": w, "traffic_bytes": t, "traffic_bytes_per_cell": t/CELLS,
        "launches_sampled": len(fetch[key])}
doc = {"how": __doc__.split("\n\n", 1)[1].replace("\n", " "), "cells": CELLS, "kernels": kernels}
json.dump(doc, open(sys.argv[3], "w"), indent=1)
for k, v in kernels.items():
    print("%-60s %8.1f MB  %7.1f B/cell" % (k[:60], v["traffic_bytes"]/1e6, v["traffic_bytes_per_cell"]))
