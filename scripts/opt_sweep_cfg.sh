#!/bin/bash
# Development probe: one configuration under engine-option overrides.  usage: opt_sweep_cfg.sh CONFIG "opts" ...
cfg=$1; shift
for o in "$@"; do
  python bench.py --config $cfg --no-cpu-baseline --long-steps 0 --steps ${STEPS:-20} $(for kv in $o; do echo --opt $kv; done) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$cfg %-40s N/s %7.2f its/s %7.1f its/N %5.1f failed %d' % ('$o', d['value'], c['fgmres_its_per_s'], c['fgmres_its']/max(1,c['newton_its']), c['failed_solves']))"
done
