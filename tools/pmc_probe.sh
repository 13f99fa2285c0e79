#!/bin/bash
# Counter passes on the pair kernel of a chosen library build (run through gpurun from the repo root):
#   tools/pmc_probe.sh <tag> [lib.so]
# One rocprofv3 --pmc pass per counter group (kernel trace only, as the gpurun rules require); per-kernel averages
# land in gpurun_out/pmc_<tag>/.  Groups that the profiler rejects (too many counters for a block) are reported.
R=$PWD
TAG=${1:-probe}
LIB=${2:-}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
[ -n "$LIB" ] && export TMDHIP_LIB=$R/$LIB
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 200 --warmup 50 --relax-steps 600 --no-cpu-baseline --no-secondary"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rm -rf /tmp/pp_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $group -d /tmp/pp_$i -- $CMD > /tmp/pp_$i.log 2>&1
  db=$(find /tmp/pp_$i -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python $R/profiles/summarize_pmc.py $db --min-us=20 list_pair build_list > $OUT/pass_$i.txt; else echo "pass $i FAILED: $group" > $OUT/pass_$i.txt; tail -5 /tmp/pp_$i.log >> $OUT/pass_$i.txt; fi
done <<'GROUPS'
TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
TA_TA_BUSY_sum TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE
SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
GROUPS
cat $OUT/pass_*.txt | grep -v "^#" | cut -c1-110
