set -x
cd $GRAFT_REPO_ROOT
R=$PWD
TAG=${1:-final}   # tools/final_profile.sh <tag>: kernel trace + PMC passes + bench lines of a round's last code commit
bash tools/profile_round.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
bash tools/pmc_c5.sh $TAG > gpurun_out/pmc_c5_$TAG.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
for i in 1 2 3 4 5 6; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${TAG}_bench_driver_flags.jsonl; done
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["roofline"]["frac"], d.get("secondary", {}).get("c5", {}).get("value"))
for l in open("gpurun_out/${TAG}_bench_driver_flags.jsonl"):
    d = json.loads(l); print("driver", d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["list"]["rebuilds_in_timed_region"], d.get("secondary", {}).get("c5", {}).get("value"))
PY
ls gpurun_out/prof_$TAG gpurun_out/pmc_c5_$TAG
