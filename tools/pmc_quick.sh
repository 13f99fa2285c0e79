#!/bin/bash
# one rocprofv3 --pmc pass on a short bench run:  tools/pmc_quick.sh <tag> <counters...>   (env LIB=path of an alternative library)
R=$PWD
tag=$1; shift
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export TMDHIP_LIB=$R/$LIB
rm -rf /tmp/pq_$tag
timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pq_$tag -- python $R/bench.py --steps 300 --warmup 50 --relax-steps 400 --no-cpu-baseline --no-secondary > /tmp/pq_$tag.log 2>&1
for f in $(find /tmp/pq_$tag -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=5 list_pair build_list md_step > $R/gpurun_out/pmc/$tag.txt; done
cat $R/gpurun_out/pmc/$tag.txt
