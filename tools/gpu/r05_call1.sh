set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_a
O=gpurun_out/r05_a
./tools/ubench/sanity 1 > $O/sanity.txt 2>&1; echo "sanity rc $?" >> $O/sanity.txt
timeout 120 ./tools/ubench/atomic_rate > $O/atomic_rate.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q -rP > $O/tests_rP.log 2>&1; tail -3 $O/tests_rP.log
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 >> $O/bench_driver_flags.jsonl; done
timeout 540 python tools/soak.py --inproc 200 --fresh 20 --out $O/soak > $O/soak_stdout.txt 2>&1
tail -5 $O/soak.log
