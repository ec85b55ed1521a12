#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "per_world or tables" > $O/pytest_tables.log 2>&1 || { tail -60 $O/pytest_tables.log; exit 1; }
tail -2 $O/pytest_tables.log
python tools/ab.py --tables --rounds 3 default@2 rs2@2 default@1 rs2@1 default@0 > $O/ab_tables2.txt 2>&1
cat $O/ab_tables2.txt
