cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_k
mkdir -p $O
for r in 1 2 3; do
  for v in base cur; do
    if [ $v = base ]; then export TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_base.so; else unset TMDHIP_LIB; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), round(d['roofline']['avg_kernel_us'],2), d['list']['rebuilds_in_timed_region'])" | tee -a $O/driver_ab.txt
  done
done
unset TMDHIP_LIB
for v in base cur; do
  if [ $v = base ]; then export TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_base.so; else unset TMDHIP_LIB; fi
  python bench.py --gpus 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('default-run $v', round(d['value'],1), round(d['ms_per_step']*1e3,2), round(d['roofline']['avg_kernel_us'],2), d['list']['rebuilds_in_timed_region'])" | tee -a $O/driver_ab.txt
done
