#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_il0.so python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_il0.log 2>&1 || { tail -40 $O/pytest_il0.log; exit 1; }
tail -2 $O/pytest_il0.log
for n in 16777216 8388608; do echo "N=$n"; python tools/ab.py --rounds 3 --envs $n --steps 200 default@2 noxcd@2 default@0 2>&1 | grep us/step; done > $O/ab_xcd.txt 2>&1
for n in 4194304 1048576; do echo "N=$n"; python tools/ab.py --rounds 2 --envs $n --steps 400 il0@2 il0noxcd@2 default@2 2>&1 | grep us/step; done >> $O/ab_xcd.txt 2>&1
cat $O/ab_xcd.txt
