#!/bin/bash
# the N > 1 code path with one rank (RCCL barrier, all-reduce, done-mask all-gather) against the plain N = 1 run
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --force-exchange > $O/bench_driver_torchrun_exchange.json 2> $O/tr1.err || { tail $O/tr1.err; exit 1; }
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --no-cpu-baseline --force-exchange > $O/bench_default_torchrun_exchange.json 2> $O/tr2.err || { tail $O/tr2.err; exit 1; }
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/d.err || { tail $O/d.err; exit 1; }
for f in bench_driver_torchrun_exchange bench_default_torchrun_exchange bench_driver; do python3 - $O/$f.json <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], r["steps"], r["config"]["done_mask_exchange"], "value %.4g ms/step %.6f launch_us %.3f" % (r["value"], r["ms_per_step"], r["roofline"]["launch_us"]), ["%.3f" % x for x in r["regions_ms"]])
PY
done
