#!/bin/bash
# rocprofv3 passes of one round on the GPU box (run through gpurun from the repo root):
# kernel trace + stats, then one --pmc pass per counter group (never combined with other trace domains).
# Summaries land in gpurun_out/prof_e/; copy what should be judged into profiles/.
set -x
R=$PWD
mkdir -p $R/gpurun_out/prof_e
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 400 --warmup 100 --relax-steps 600 --no-cpu-baseline"
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- $CMD > /tmp/p_stats.log 2>&1
for f in $(find /tmp/p_stats -name "*_results.db"); do python $R/profiles/summarize_rocpd.py $f > $R/gpurun_out/prof_e/kernel_stats.csv; done
timeout 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -- $CMD > /tmp/p_fetch.log 2>&1
for f in $(find /tmp/p_fetch -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=20 list_pair build_list md_step > $R/gpurun_out/prof_e/pmc_fetch.txt; done
timeout 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/p_write -- $CMD > /tmp/p_write.log 2>&1
for f in $(find /tmp/p_write -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=20 list_pair build_list md_step > $R/gpurun_out/prof_e/pmc_write.txt; done
timeout 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/p_sq -- $CMD > /tmp/p_sq.log 2>&1
for f in $(find /tmp/p_sq -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=8 list_pair build_list md_step > $R/gpurun_out/prof_e/pmc_sq.txt; done
for f in /tmp/p_stats.log /tmp/p_fetch.log; do tail -n 3 $f | cut -c1-200; done
ls -la $R/gpurun_out/prof_e
