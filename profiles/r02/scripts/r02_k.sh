#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python3 bench.py --no-cpu-baseline --extras --per-world-tables > $O/bench_extras.json 2> $O/bench_extras.err || { tail -20 $O/bench_extras.err; exit 1; }
cat $O/bench_extras.json
