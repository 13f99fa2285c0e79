cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_m
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_integrator.py tests/test_gpu_driver.py -x -q -m gpu -k "water291 or ala2 or thrombin or vmap or switch or replica or tiny or minimum_image or auto_falls or small or md_run or external or minimi or conf" > $O/tests_allpairs.log 2>&1; tail -3 $O/tests_allpairs.log
for r in 1 2; do
for v in base cur; do
  if [ $v = base ]; then export TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_base.so; else unset TMDHIP_LIB; fi
  for a in "ala2 1 100 20" "ala2 1 10 100" "water291 2 100 20" "water291 16 100 10" "ala2 16 100 10"; do echo "$v: $(python tools/small_calls.py $a 2>/dev/null | tail -1)" | tee -a $O/small_ab.txt; done
done
done
