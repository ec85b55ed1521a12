#!/bin/bash
set -e
mkdir -p gpurun_out/r02t
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02t/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r02t/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r02t/pytest_gpu.log
timeout -k 10 600 python tools/ab.py --tables --rounds 3 ref@0 default@0 ref@1 default@1 ref@2 default@2 > gpurun_out/r02t/ab_tables.txt 2>&1 || { tail -30 gpurun_out/r02t/ab_tables.txt; exit 1; }
cat gpurun_out/r02t/ab_tables.txt
timeout -k 10 600 python tools/ab.py --rounds 2 ref@0 default@0 ref@1 default@1 ref@2 default@2 > gpurun_out/r02t/ab_shared.txt 2>&1 || { tail -30 gpurun_out/r02t/ab_shared.txt; exit 1; }
cat gpurun_out/r02t/ab_shared.txt
