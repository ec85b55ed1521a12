#!/bin/bash
# round 4, first GPU contact of the slim-argument next-step kernel: the GPU suite, A/B against round 3's library, SQ counters
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04/contact1; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
python3 tools/ab.py --rounds 4 r03@2 default@2 r03@1 default@1 r03@0 default@0 > $O/ab_262144.txt 2>&1; cat $O/ab_262144.txt
python3 tools/ab.py --rounds 2 --envs 16777216 --steps 200 r03@2 default@2 > $O/ab_16m.txt 2>&1; cat $O/ab_16m.txt
bash tools/r04/pmc_sq.sh r04/contact1/pmc_sq > $O/pmc_sq.txt 2>&1; cat $O/pmc_sq.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; r=json.load(open('$O/bench_driver.json')); print('driver flags: launch_us %.3f frac %.4f ms/step %.5f' % (r['roofline']['launch_us'], r['roofline']['frac'], r['ms_per_step']))"
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; r=json.load(open('$O/bench_default.json')); print('defaults: launch_us %.3f frac %.4f ms/step %.5f' % (r['roofline']['launch_us'], r['roofline']['frac'], r['ms_per_step']))"
