#!/bin/bash
# Round-end measurement set on the one-GPU box; results under gpurun_out/final/ (copied to profiles/ by hand).
#   bash scripts/measure_all.sh a   -> GPU test suite, bench (default and driver command)
#   bash scripts/measure_all.sh b   -> other configurations, rocprofv3 kernel stats, PMC traffic passes
#   bash scripts/measure_all.sh c   -> config 5, one slab (9 M cells: several minutes of dt ramp)
set -e -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
case "$1" in
a)
  python -m pytest tests -m gpu -q -x -v > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
  tail -3 $OUT/pytest_gpu.log
  python bench.py --steps 8 --warmup 2 > $OUT/bench_short_window.json 2> $OUT/bench_short_window.err     # (a short window, with its failed solve)
  python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
  ;;
b)
  for c in c1 c2 c3; do
    python bench.py --config $c --long-steps 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --long-steps 0 \
      > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
  find $OUT/kt -name '*kernel_trace.csv' -delete
  TP_GRAPH=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/pmc_probe.py > $OUT/pmc_fetch.log 2>&1
  TP_GRAPH=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/pmc_probe.py > $OUT/pmc_write.log 2>&1
  python3 scripts/pmc_summarize.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json
  find $OUT/pmc_fetch $OUT/pmc_write -name '*kernel_trace.csv' -delete
  du -sh $OUT
  ;;
c)
  python bench.py --config c5slab --no-cpu-baseline --steps 6 --warmup 2 --long-steps 0 > $OUT/bench_c5slab.json 2> $OUT/bench_c5slab.err
  ;;
esac
