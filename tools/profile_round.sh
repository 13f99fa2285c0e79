#!/bin/bash
# rocprofv3 passes of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>          e.g. r02_a
# kernel trace + stats, then one --pmc pass per counter group (never combined with other trace domains).
# Summaries land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
set -x
R=$PWD
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write /tmp/p_sq /tmp/p_tcp  # (a second pass on the same box must not find the first one's databases)
# BENCH_EXTRA: further bench.py options, e.g. BENCH_EXTRA="--switch-dist 7.5" profiles the SWITCH variant of the launch
CMD="python $R/bench.py --steps 400 --warmup 100 --relax-steps 600 --no-cpu-baseline --no-secondary ${BENCH_EXTRA:-}"
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- $CMD > /tmp/p_stats.log 2>&1
for f in $(find /tmp/p_stats -name "*_results.db"); do python $R/profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
timeout 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -- $CMD > /tmp/p_fetch.log 2>&1
for f in $(find /tmp/p_fetch -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=20 list_pair build_list md_step > $OUT/pmc_fetch.txt; done
timeout 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/p_write -- $CMD > /tmp/p_write.log 2>&1
for f in $(find /tmp/p_write -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=20 list_pair build_list md_step > $OUT/pmc_write.txt; done
timeout 280 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/p_sq -- $CMD > /tmp/p_sq.log 2>&1
for f in $(find /tmp/p_sq -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=8 list_pair build_list md_step > $OUT/pmc_sq.txt; done
timeout 280 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS TA_BUSY_avr -d /tmp/p_tcp -- $CMD > /tmp/p_tcp.log 2>&1
for f in $(find /tmp/p_tcp -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=20 list_pair build_list > $OUT/pmc_tcp.txt; done
git -C $R rev-parse HEAD > $OUT/commit.txt 2>/dev/null || true
for f in /tmp/p_stats.log /tmp/p_fetch.log /tmp/p_tcp.log; do tail -n 2 $f | cut -c1-200; done
ls -la $OUT
