#!/bin/bash
set -e
mkdir -p gpurun_out/r02t
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "per_world or tables" > gpurun_out/r02t/pytest_tables2.log 2>&1 || { tail -40 gpurun_out/r02t/pytest_tables2.log; exit 1; }
tail -2 gpurun_out/r02t/pytest_tables2.log
timeout -k 10 600 python tools/tables_fused_time.py > gpurun_out/r02t/tables_fused.txt 2>&1 || { tail -30 gpurun_out/r02t/tables_fused.txt; exit 1; }
cat gpurun_out/r02t/tables_fused.txt
