#!/bin/bash
# round 2, call b: parity suite with the pair-cooperative re-seeding + A/B against the round-1 routine
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 3 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
cat $O/bench_driver.json
python tools/ab.py --rounds 3 default@2 r1@2 default@1 r1@1 > $O/ab_reseed.txt 2>&1
cat $O/ab_reseed.txt
python tools/ab.py --rounds 2 --envs 16777216 --steps 200 default@2 r1@2 > $O/ab_reseed_16m.txt 2>&1
cat $O/ab_reseed_16m.txt
