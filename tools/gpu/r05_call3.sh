set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in base default dbg1 dbg2; do
  if [ $v = default ]; then unset TMDHIP_LIB; else export TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_$v.so; fi
  rm -rf /tmp/tb_$v
  timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/tb_$v -- python $R/tools/time_build.py 20 > $O/tb_$v.log 2>&1
  grep TIMEBUILD $O/tb_$v.log
  for f in $(find /tmp/tb_$v -name "*_results.db"); do python $R/profiles/summarize_rocpd.py $f | grep -i "build_list\|scan_place\|bin_members" | cut -c1-60,120-200 > $O/tb_$v.csv; done
  cat $O/tb_$v.csv
done
