#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
for r in 1 2 3; do
  echo "round $r"
  python ref_r01/tools/ab.py --rounds 1 default@2 default@1 default@0 2>&1 | sed 's/^/ref   /'
  python tools/ab.py --rounds 1 default@2 default@1 default@0 2>&1 | sed 's/^/new   /'
done > $O/ab_vs_ref.txt 2>&1
cat $O/ab_vs_ref.txt
python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_il0.so python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_il0.log 2>&1 || { tail -40 $O/pytest_il0.log; exit 1; }
tail -2 $O/pytest_il0.log
