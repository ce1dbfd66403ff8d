#!/bin/bash
# Development probe: C4 through the time loop (20 timed steps) under engine-option overrides; one summary line per run.
for o in "$@"; do
  python bench.py --no-cpu-baseline --long-steps 0 --steps ${STEPS:-20} $(for kv in $o; do echo --opt $kv; done) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-48s N/s %6.2f its/s %6.1f its/N %5.1f failed %d vcycle %.3f pc_apply %.3f' % ('$o', d['value'], c['fgmres_its_per_s'], c['fgmres_its']/max(1,c['newton_its']), c['failed_solves'], c['kernels_ms']['amg_vcycle_ms'], c['kernels_ms']['pc_apply_ms']))"
done
