cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_l
for a in "ala2 1 10" "ala2 1 100" "ala2 1 1" "water291 2 10" "ala2 16 10"; do python tools/small_overhead.py $a 2>/dev/null | tail -1 | tee -a gpurun_out/r05_l/small_overhead.txt; done
