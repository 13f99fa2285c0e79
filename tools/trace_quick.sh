#!/bin/bash
# kernel trace + stats of a short bench run: tools/trace_quick.sh <tag> [bench args]; summary -> gpurun_out/trace/<tag>.csv
R=$PWD
tag=$1; shift
mkdir -p $R/gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tq_$tag
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tq_$tag -- python $R/bench.py --steps 400 --warmup 50 --relax-steps 400 --no-cpu-baseline --no-secondary "$@" > /tmp/tq_$tag.log 2>&1
for f in $(find /tmp/tq_$tag -name "*_results.db"); do python $R/profiles/summarize_rocpd.py $f > $R/gpurun_out/trace/$tag.csv; done
head -14 $R/gpurun_out/trace/$tag.csv | cut -c1-160
