#!/bin/bash
# final bench lines of round 2 with the final bench.py (kernels unchanged since the rocprofv3 passes: same library hash)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
sha256sum aquaticgymenv_amd/lib/libaqua_hip.so | cut -c1-16 > $O/library_sha16.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail $O/bench_driver.err; exit 1; }
python3 bench.py --extras --per-world-tables > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
python3 bench.py --envs 16777216 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_n16m.json 2> $O/bench_n16m.err || { tail $O/bench_n16m.err; exit 1; }
python3 bench.py --continuous --no-cpu-baseline > $O/bench_continuous.json 2> $O/bench_continuous.err || { tail $O/bench_continuous.err; exit 1; }
python3 bench.py --reset-mode 1 --no-cpu-baseline > $O/bench_same_step.json 2> $O/bench_same_step.err || { tail $O/bench_same_step.err; exit 1; }
python3 bench.py --no-auto-reset --no-cpu-baseline > $O/bench_no_restart.json 2> $O/bench_no_restart.err || { tail $O/bench_no_restart.err; exit 1; }
python3 bench.py --envs 4096 --no-obstacles --no-cpu-baseline > $O/bench_n4096_noobst.json 2> $O/bench_n4096.err || { tail $O/bench_n4096.err; exit 1; }
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --force-exchange > $O/bench_driver_torchrun_exchange.json 2> $O/bench_torchrun.err || { tail $O/bench_torchrun.err; exit 1; }
for f in driver default n16m continuous same_step no_restart n4096_noobst driver_torchrun_exchange; do python3 - $O/bench_$f.json <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], "value %.4g  ms/step %.6f  launch_us %.3f (%s)  frac %.4f  traffic %s" % (r["value"], r["ms_per_step"], r["roofline"]["launch_us"], r["roofline"]["launch_us_events"], r["roofline"]["frac"], r["roofline"]["traffic"]))
PY
done
