#!/bin/bash
# kernel trace + the PMC passes of config C5 on one GPU: tools/pmc_c5.sh <tag>  -> gpurun_out/pmc_c5_<tag>/
R=$PWD
tag=${1:-c5}
OUT=$R/gpurun_out/pmc_c5_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5_stats /tmp/c5_fetch /tmp/c5_write  # (a box may be handed out again with its /tmp)
CMD="python $R/bench.py --config c5 --steps 120 --warmup 20 --no-cpu-baseline"
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/c5_stats -- $CMD > /tmp/c5_stats.log 2>&1
for f in $(find /tmp/c5_stats -name "*_results.db"); do python $R/profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
timeout 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/c5_fetch -- $CMD > /tmp/c5_fetch.log 2>&1
for f in $(find /tmp/c5_fetch -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=20 list_pair build_list md_step > $OUT/pmc_fetch.txt; done
timeout 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES -d /tmp/c5_write -- $CMD > /tmp/c5_write.log 2>&1
for f in $(find /tmp/c5_write -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=20 list_pair build_list md_step > $OUT/pmc_write.txt; done
head -6 $OUT/kernel_stats.csv | cut -c1-70,110-170; cat $OUT/pmc_fetch.txt $OUT/pmc_write.txt | head -30
