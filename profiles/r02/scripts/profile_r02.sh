#!/bin/bash
# rocprofv3 evidence of round 2, all on ONE box: the driver's command, the default bench line, the streaming size,
# continuous actions; then bench lines of the same box without the tracer
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
T=${1:-r02}
bash tools/profile_round.sh $T/k20 --steps 20 --warmup 5 || exit 1
bash tools/profile_round.sh $T/n262144 || exit 1
bash tools/profile_round.sh $T/n16m --envs 16777216 --steps 100 --warmup 20 || exit 1
bash tools/profile_round.sh $T/n262144_cont --continuous --steps 500 --warmup 100 || exit 1
O=$R/gpurun_out/$T
python3 tools/make_traffic.py $O n262144_disc_k8=n262144 n16777216_disc_k8=n16m n262144_cont_k8=n262144_cont > $O/traffic.log 2>&1 || { cat $O/traffic.log; exit 1; }
cp profiles/traffic.json $O/traffic.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail $O/bench_driver.err; exit 1; }
python3 bench.py --extras --per-world-tables > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
python3 bench.py --envs 16777216 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_n16m.json 2> $O/bench_n16m.err || { tail $O/bench_n16m.err; exit 1; }
python3 bench.py --continuous --no-cpu-baseline > $O/bench_continuous.json 2> $O/bench_continuous.err || { tail $O/bench_continuous.err; exit 1; }
python3 bench.py --reset-mode 1 --no-cpu-baseline > $O/bench_same_step.json 2> $O/bench_same_step.err || { tail $O/bench_same_step.err; exit 1; }
python3 bench.py --no-auto-reset --no-cpu-baseline > $O/bench_no_restart.json 2> $O/bench_no_restart.err || { tail $O/bench_no_restart.err; exit 1; }
python3 bench.py --envs 4096 --no-obstacles --no-cpu-baseline > $O/bench_n4096_noobst.json 2> $O/bench_n4096.err || { tail $O/bench_n4096.err; exit 1; }
for f in driver default n16m continuous same_step no_restart n4096_noobst; do python3 - $O/bench_$f.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value %.4g  ms/step %.6f  launch_us %.3f (%s)  frac %.4f  traffic %s" % (r["value"], r["ms_per_step"], r["roofline"]["launch_us"], r["roofline"]["launch_us_events"], r["roofline"]["frac"], r["roofline"]["traffic"]))
PY
done
