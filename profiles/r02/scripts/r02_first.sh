#!/bin/bash
# round 2, first GPU call: the bench harness as the driver runs it + the rocprofv3 evidence of the same commands
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
python -m pytest tests/test_00_bench_child.py -m gpu -x -q > $O/pytest_bench_child.log 2>&1 || { tail -30 $O/pytest_bench_child.log; exit 1; }
tail -3 $O/pytest_bench_child.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
cat $O/bench_driver.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
bash tools/profile_round.sh r02a/k20 --steps 20 --warmup 5 || exit 1
bash tools/profile_round.sh r02a/n262144 || exit 1
