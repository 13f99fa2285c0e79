#!/bin/bash
# A/B of the domain-decomposition loop on one GPU (one brick exchanging its periodic images with itself over RCCL):
#   tools/dd_ab.sh <rounds> "ENV=val ..." ...     ("" = defaults); prints us/step, migrations and their total time
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for s in "$@"; do
    env $s DD_TIME_MIGRATIONS=1 timeout 300 python tools/bench_dd.py --nside ${NSIDE:-50} --steps ${STEPS:-2000} 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
m=d['migrations_in_timed_region']
print('%-40s %7.2f us/step  %7.1f ns/day  migrations %2d  %6.2f ms each  (%5.2f us/step)  steady %6.2f us/step' % ('$s' or 'default', d['us_per_step'], d['ns_per_day'], m, d['migration_ms_total']/max(m,1), d['migration_ms_total']*1e3/d['steps'], d['us_per_step']-d['migration_ms_total']*1e3/d['steps']))
" || echo "$s FAILED"
  done
done
