#!/bin/bash
# A/B timing of differently built libraries on the GPU box: tools/ab.sh lib1.so lib2.so ... (bench, no CPU baseline)
for lib in "$@"; do
  echo "== $lib"
  TMDHIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
l=d['list']
print('ms/step %.4f  ns/day %.1f  pair_us %.2f  rebuilds %s  entries %s  T %.1f Epot %.1f' % (d['ms_per_step'], d['value'], d['roofline']['avg_kernel_us'], l['rebuilds_in_timed_region'], l['entries'], d['temperature_K'][0], d['epot_kcal_mol'][0]))"
done
