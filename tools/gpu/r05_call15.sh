cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_o
mkdir -p $O
TMDHIP_DEBUG_POISON=1 timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_poison.log 2>&1; grep -n "passed\|failed" $O/tests_poison.log | tail -2
timeout 300 python tools/soak.py --inproc 100 --fresh 10 --out $O/soak > $O/soak_stdout.txt 2>&1; tail -3 $O/soak.log
for a in "ala2 1 100 20 f64" "ala2 1 10 100 f64" "ala2 16 100 10 f64" "thrombin 1 50 10 f32" "thrombin 1 50 10 f64" "water291 2 10 100 f32"; do python tools/small_calls.py $a 2>/dev/null | tail -1 | tee -a $O/small_more.txt; done
TMDHIP_ALLPAIRS_WAVES=1024 python tools/small_calls.py thrombin 1 50 10 f32 2>/dev/null | tail -1 | sed 's/^/waves 1024: /' | tee -a $O/small_more.txt
python tools/time_small.py 2>/dev/null | tee $O/time_small.txt
