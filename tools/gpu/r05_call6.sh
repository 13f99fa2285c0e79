set -x
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r05_f
mkdir -p $O
python tools/ab_pair.py torchmd_amd/lib/exp/libtmdhip_base.so default torchmd_amd/lib/exp/libtmdhip_w7.so --rounds 2 > $O/ab_prefilter2.txt 2>&1; tail -5 $O/ab_prefilter2.txt
cd /tmp && export TMPDIR=/tmp
for v in default w7; do
  if [ $v = default ]; then unset TMDHIP_LIB; else export TMDHIP_LIB=$R/torchmd_amd/lib/exp/libtmdhip_$v.so; fi
  rm -rf /tmp/pm_$v
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pm_$v -- python $R/tools/time_build.py 12 > $O/pm_$v.log 2>&1
  for f in $(find /tmp/pm_$v -name "*_results.db"); do python $R/profiles/summarize_pmc.py $f --min-us=50 build_list > $O/pmc_$v.txt; done
  cat $O/pmc_$v.txt
done
