#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 2 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
cat $O/bench_driver.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
python tools/ab.py --rounds 2 --envs 16777216 --steps 200 default@2 default@0:1 default@0:2 default@0:4 default@1:1 default@1:2 > $O/ab_16m_vec.txt 2>&1
cat $O/ab_16m_vec.txt
python tools/ab.py --rounds 2 --envs 4194304 --steps 400 default@2 default@0:1 default@0:2 default@0:4 > $O/ab_4m_vec.txt 2>&1
cat $O/ab_4m_vec.txt
