cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_n
mkdir -p $O
for w in 1024 2048 4096 8192 0; do
  export TMDHIP_ALLPAIRS_WAVES=$w
  for a in "ala2 16 100 10" "water291 16 100 10" "water291 64 100 5" "ala2 1 100 20"; do echo "waves $w: $(python tools/small_calls.py $a 2>/dev/null | tail -1)" | tee -a $O/waves_sweep.txt; done
done
unset TMDHIP_ALLPAIRS_WAVES
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc; timeout 100 rocprofv3 --kernel-trace -d /tmp/sc -- python $R/tools/small_calls.py ala2 16 10 20 > $O/small_ala2x16.txt 2>&1
for f in $(find /tmp/sc -name "*_results.db"); do python $R/tools/call_timeline.py $f 2 > $O/timeline_ala2x16.txt; done
head -12 $O/timeline_ala2x16.txt
