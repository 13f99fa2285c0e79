set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_h
mkdir -p $O
for v in base w7 default; do
  if [ $v = default ]; then unset TMDHIP_LIB; else export TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_$v.so; fi
  timeout 100 python tools/build_timeline.py > $O/timeline_$v.txt 2>&1
  grep -v "^xcc\|entry-time" $O/timeline_$v.txt | tail -12
done
