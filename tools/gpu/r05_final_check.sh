#!/bin/bash
# final validation of the round: full GPU suite, smoke, fp64 / small-system numbers, the driver's command
mkdir -p gpurun_out/r05_final
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05_final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee gpurun_out/r05_final/smoke.txt
timeout 300 python tools/time_fp64.py 2>&1 | tail -1 | tee gpurun_out/r05_final/fp64.txt
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r05_final/bench_driver_flags.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/r05_final/bench_driver_flags.jsonl"):
    d = json.loads(l); print("driver", round(d["value"],1), round(d["ms_per_step"]*1e3,2), round(d["roofline"]["avg_kernel_us"],2), d["list"]["rebuilds_in_timed_region"], d.get("secondary", {}).get("c5", {}).get("value"), d["cpu_baseline"]["value"])
PY
