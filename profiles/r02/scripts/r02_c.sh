#!/bin/bash
# round 2, call c: A/B of the pair-cooperative re-seeding (registers fixed) against the round-1 routine
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python tools/ab.py --rounds 3 default@2 r1@2 default@1 r1@1 default@0 > $O/ab_reseed.txt 2>&1
cat $O/ab_reseed.txt
python tools/ab.py --rounds 2 --envs 16777216 --steps 200 default@2 r1@2 > $O/ab_reseed_16m.txt 2>&1
cat $O/ab_reseed_16m.txt
