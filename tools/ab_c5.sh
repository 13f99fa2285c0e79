#!/bin/bash
# A/B of environment settings on the 10^6-atom LJ box (bench.py --config c5, one GPU): tools/ab_c5.sh <rounds> "ENV=val" ...
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for s in "$@"; do
    env $s python bench.py --config c5 --no-cpu-baseline --steps ${STEPS:-600} --warmup ${WARMUP:-100} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-60s %8.1f ns/day %7.2f us/step  pair %6.2f us' % ('$s' or 'default', d['value'], d['ms_per_step']*1e3, r['avg_kernel_us']))
" || echo "$s FAILED"
  done
done
