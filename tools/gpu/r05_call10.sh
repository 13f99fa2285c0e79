set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_j
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -rP > $O/tests_rP.log 2>&1; tail -3 $O/tests_rP.log
for r in 1 2; do
  for v in 0 1; do
    echo "== TMDHIP_FUSED_FINAL=$v" >> $O/short_call_ab.txt
    TMDHIP_FUSED_FINAL=$v python tools/short_call.py 20 40 2>/dev/null >> $O/short_call_ab.txt
  done
done
cat $O/short_call_ab.txt
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 >> $O/bench_driver_flags.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/r05_j/bench_driver_flags.jsonl"):
    d = json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["list"]["rebuilds_in_timed_region"])
PY
