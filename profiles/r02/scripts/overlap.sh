#!/bin/bash
# stream-overlap probes (tools/overlap_probe.py, tools/overlap_roles.py)
set -e
mkdir -p gpurun_out/r02o
timeout -k 10 300 python tools/overlap_roles.py 262144 100 20 > gpurun_out/r02o/overlap_roles.txt 2>&1 || { tail -30 gpurun_out/r02o/overlap_roles.txt; exit 1; }
cat gpurun_out/r02o/overlap_roles.txt
