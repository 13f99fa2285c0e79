set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_i
mkdir -p $O
python tools/ab_pair.py torchmd_amd/lib/exp/libtmdhip_base.so default --rounds 2 > $O/ab_prefilter4.txt 2>&1; tail -4 $O/ab_prefilter4.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc; timeout 100 rocprofv3 --kernel-trace -d /tmp/sc -- python $R/tools/small_calls.py ala2 1 10 40 > $O/small_ala2.txt 2>&1
for f in $(find /tmp/sc -name "*_results.db"); do python $R/tools/call_timeline.py $f 2 > $O/timeline_ala2.txt; done
cat $O/small_ala2.txt | tail -1; cat $O/timeline_ala2.txt
rm -rf /tmp/sc2; timeout 100 rocprofv3 --kernel-trace -d /tmp/sc2 -- python $R/tools/small_calls.py water291 16 10 40 > $O/small_w16.txt 2>&1
for f in $(find /tmp/sc2 -name "*_results.db"); do python $R/tools/call_timeline.py $f 2 > $O/timeline_w16.txt; done
tail -1 $O/small_w16.txt; cat $O/timeline_w16.txt
cd $R
python tools/small_calls.py ala2 1 10 200 | tail -1
python tools/small_calls.py ala2 1 100 20 | tail -1
python tools/small_calls.py ala2 16 10 100 | tail -1
