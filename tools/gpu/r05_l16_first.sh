#!/bin/bash
# first GPU run of the 16-bit list: its tests, then C5 and the 786 432-atom water box with the format on / off
mkdir -p gpurun_out/r05_l16
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sixteen_bit or streamed_list" 2>&1 | tail -15
echo "== C5"
STEPS=400 WARMUP=100 timeout 900 bash tools/ab_c5.sh 2 "TMDHIP_LIST16=0" "TMDHIP_LIST16=1" 2>&1 | tee gpurun_out/r05_l16/c5_ab.txt
echo "== water 786k"
BENCH_ARGS="--nside 64 --steps 300 --warmup 100" timeout 900 bash tools/ab_env.sh 1 "TMDHIP_LIST16=0" "TMDHIP_LIST16=1" 2>&1 | tee gpurun_out/r05_l16/water786k_ab.txt
