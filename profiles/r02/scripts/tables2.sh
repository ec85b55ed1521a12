#!/bin/bash
# per-world tables: restart inside the tile with the rows handed over through LDS -- tests, then A/B against the round's earlier kernels
set -e
mkdir -p gpurun_out/r02t
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "per_world or tables" > gpurun_out/r02t/pytest_tables.log 2>&1 || { tail -40 gpurun_out/r02t/pytest_tables.log; exit 1; }
tail -2 gpurun_out/r02t/pytest_tables.log
timeout -k 10 600 python tools/ab.py --tables --rounds 2 ref@0 default@0 ref@1 default@1 ref@2 default@2 > gpurun_out/r02t/ab_tables.txt 2>&1 || { tail -30 gpurun_out/r02t/ab_tables.txt; exit 1; }
cat gpurun_out/r02t/ab_tables.txt
