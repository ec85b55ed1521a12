#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
for v in stamps stamps_r1 stamps_p; do echo $v; AQUA_HIP_LIB=aquaticgymenv_amd/lib/variants/libaqua_hip_$v.so python tools/reseed_bench.py; done > $O/reseed_bench.txt 2>&1
cat $O/reseed_bench.txt
python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python tools/ab.py --rounds 2 default@2 r1@2 lds@2 nm@2 default@1 r1@1 > $O/ab.txt 2>&1
cat $O/ab.txt
