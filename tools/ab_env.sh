#!/bin/bash
# A/B of environment settings on one box: every argument is one setting ("VAR=val VAR2=val2", "" = defaults); the
# default bench (2 000 steps, no CPU baseline, no secondary leg) runs once per setting and round, alternating.
#   tools/ab_env.sh 2 "" "TMDHIP_VSKIN=0" "TMDHIP_CHAIN_SKIP=0"
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for s in "$@"; do
    env $s python bench.py --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['list']; r=d['roofline']
print('%-44s %8.1f ns/day %7.2f us/step  pair %6.2f us (%d timed)  rebuilds %3d (%.1f steps)  skipped %d  T %.1f' % ('$s' or 'default', d['value'], d['ms_per_step']*1e3, r['avg_kernel_us'], r['launches_timed'], l['rebuilds_in_timed_region'], l['steps_per_rebuild'] or 0, l['rebuild_chains_left_out'], d['temperature_K'][0]))
" || echo "$s FAILED"
  done
done
