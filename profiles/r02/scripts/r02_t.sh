#!/bin/bash
# the one-sequence placement specification: parity (both grid layouts), then the restart modes against the round's first kernels
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_il0.so python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_il0.log 2>&1 || { tail -60 $O/pytest_il0.log; exit 1; }
tail -2 $O/pytest_il0.log
for r in 1 2 3; do
  python ref_r01/tools/ab.py --rounds 1 default@2 default@1 2>&1 | grep us/step | sed 's/^/ref   /'
  python tools/ab.py --rounds 1 default@2 default@1 2>&1 | grep us/step | sed 's/^/new   /'
done > $O/ab_placement.txt 2>&1
python tools/ab.py --rounds 2 --envs 16777216 --steps 200 default@2 >> $O/ab_placement.txt 2>&1
python tools/ab.py --tables --rounds 2 default@2 default@1 >> $O/ab_placement.txt 2>&1
cat $O/ab_placement.txt
