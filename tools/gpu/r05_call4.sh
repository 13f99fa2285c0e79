set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_d
mkdir -p $O
python tools/ab_pair.py torchmd_amd/lib/exp/libtmdhip_base.so default torchmd_amd/lib/exp/libtmdhip_w7.so --rounds 2 > $O/ab_prefilter.txt 2>&1; tail -5 $O/ab_prefilter.txt
cd /tmp && export TMPDIR=/tmp
for v in default w7; do
  if [ $v = default ]; then unset TMDHIP_LIB; else export TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_$v.so; fi
  rm -rf /tmp/tb_$v
  timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/tb_$v -- python $R/tools/time_build.py 20 > $O/tb_$v.log 2>&1
  grep TIMEBUILD $O/tb_$v.log
  for f in $(find /tmp/tb_$v -name "*_results.db"); do python - $f <<'PY' > $O/tb_$v.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for pat in ("build_list", "scan_place", "bin_members", "list_pair"):
    rows = [r[0] / 1e3 for r in db.execute("select end - start from kernels where name like ?", (f"%{pat}%",))]
    big = [x for x in rows if x > (50 if pat == "build_list" else 6 if pat != "list_pair" else 0)]
    if big:
        print(f"{pat}: {len(big)} working launches, mean {sum(big) / len(big):.1f} us, min {min(big):.1f}, max {max(big):.1f}  (of {len(rows)} launches)")
PY
  done
  cat $O/tb_$v.txt
done
unset TMDHIP_LIB
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; tail -3 $O/tests_parity.log
