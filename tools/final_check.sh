#!/bin/bash
# Final validation of a round on a GPU box: full GPU suite, smoke, the bench lines of record (default run + six runs of the
# driver's command), the kernel trace of the bench.      gpurun --timeout 3000 -- 'bash tools/final_check.sh r06'
R=$PWD
TAG=${1:-final}
O=$R/gpurun_out/${TAG}_final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rm -f $O/bench_driver_flags.jsonl
for i in 1 2 3 4 5 6; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_driver_flags.jsonl; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats  # (a box may be handed out again with its /tmp)
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- python $R/bench.py --steps 400 --warmup 100 --relax-steps 600 --no-cpu-baseline --no-secondary > /tmp/p_stats.log 2>&1
for f in $(find /tmp/p_stats -name "*_results.db"); do python $R/profiles/summarize_rocpd.py $f > $O/kernel_stats.csv; done
cd $R
export TAG_FINAL=${TAG}_final
python - <<'PY'
import json
d = json.loads(open("gpurun_out/" + __import__("os").environ.get("TAG_FINAL", "final_final") + "/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"],1), round(d["ms_per_step"]*1e3,2), round(d["roofline"]["avg_kernel_us"],2), round(d["roofline"]["frac"],4), d.get("secondary", {}).get("c5", {}).get("value"), d.get("secondary", {}).get("c3_switch", {}).get("value"))
for l in open("gpurun_out/" + __import__("os").environ.get("TAG_FINAL", "final_final") + "/bench_driver_flags.jsonl"):
    d = json.loads(l); print("driver", round(d["value"],1), round(d["ms_per_step"]*1e3,2), round(d["roofline"]["avg_kernel_us"],2), d["list"]["rebuilds_in_timed_region"], d.get("secondary", {}).get("c5", {}).get("value"))
PY
head -5 $O/kernel_stats.csv | cut -c1-160
