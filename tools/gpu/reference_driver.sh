#!/bin/bash
# The reference's own torchmd/run.py on the MI355X classes (tests/test_gpu_reference_driver.py), on a GPU box.
# Run from the build container:   tools/gpu/reference_driver.sh
# The GPU box has no /root/reference: an UNTRACKED scratch copy of the reference's package and of the two fixture
# directories travels with this one gpurun snapshot (.scratch/ is git-ignored) and is deleted again afterwards, so
# that no later snapshot and no commit carries reference source.  The log is what gets committed (profiles/).
set -e
cd "$(dirname "$0")/../.."
REF=${TORCHMD_REFERENCE_ROOT:-/root/reference}
S=.scratch/reference
rm -rf $S && mkdir -p $S/tests/data $S/tests/prod_alanine_dipeptide_amber
cp -r $REF/torchmd $S/torchmd
cp -r $REF/tests/water $S/tests/water
cp -r $REF/tests/data/prod_alanine_dipeptide_amber $S/tests/data/prod_alanine_dipeptide_amber
cp $REF/tests/prod_alanine_dipeptide_amber/conf.yaml $S/tests/prod_alanine_dipeptide_amber/conf.yaml
find $S -name __pycache__ -prune -exec rm -rf {} +
trap 'rm -rf .scratch/reference' EXIT
gpurun --timeout 900 -- 'mkdir -p gpurun_out/r06; python -m pytest tests/test_gpu_reference_driver.py -m gpu -q -s -rs 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/reference_run_py.log | tail -15'
