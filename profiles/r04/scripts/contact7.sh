#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/contact7; mkdir -p $O
python3 tools/ab.py --rounds 3 cur@2 prio0@2 prio1@2 rlast@2 > $O/ab_prio_and_order.txt 2>&1; cat $O/ab_prio_and_order.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; python3 -c "
import json; r=json.load(open('$O/bench_driver.json')); print('driver flags: launch_us %.3f frac %.4f ms/step %.5f' % (r['roofline']['launch_us'], r['roofline']['frac'], r['ms_per_step'])); print(r['roofline']['launch_us_regions'], r['regions_ms'], r['config']['timed_graph_first_replay'])"
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; r=json.load(open('$O/bench_default.json')); print('defaults: launch_us %.3f frac %.4f ms/step %.5f' % (r['roofline']['launch_us'], r['roofline']['frac'], r['ms_per_step'])); print(r['sanity'])"
bash tools/profile_round.sh r04/contact7/k20 --steps 20 --warmup 5 2>&1 | tail -12
bash tools/profile_round.sh r04/contact7/n262144 2>&1 | tail -12
