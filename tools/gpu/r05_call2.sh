set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -rP > $O/tests_rP.log 2>&1; tail -3 $O/tests_rP.log
python tools/ab_pair.py torchmd_amd/lib/exp/libtmdhip_base.so default --rounds 2 > $O/ab_build.txt 2>&1; tail -4 $O/ab_build.txt
bash tools/trace_quick.sh r05_b_new > $O/trace_new.txt 2>&1
TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_base.so bash tools/trace_quick.sh r05_b_base > $O/trace_base.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct; timeout 200 rocprofv3 --kernel-trace -d /tmp/ct -- python $R/tools/short_call.py 20 8 > $O/short_call_8.txt 2>&1
for f in $(find /tmp/ct -name "*_results.db"); do python $R/tools/call_timeline.py $f 2 > $O/call_timeline.txt; python $R/tools/call_timeline.py $f 3 > $O/call_timeline_3.txt; done
cd $R
python tools/short_call.py 20 40 > $O/short_call_40.txt 2>&1
tail -4 $O/short_call_40.txt
