#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02s
mkdir -p $O
cd $R
python3 bench.py --no-cpu-baseline > $O/plain.json 2> $O/e0.err
python3 bench.py --no-cpu-baseline --force-exchange > $O/plain_exchange.json 2> $O/e1.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/dist.json 2> $O/e2.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --no-cpu-baseline --force-exchange > $O/dist_exchange.json 2> $O/e3.err
for f in plain plain_exchange dist dist_exchange; do python3 - $O/$f.json <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.6f launch_us %.3f" % (r["value"], r["ms_per_step"], r["roofline"]["launch_us"]), ["%.3f" % x for x in r["regions_ms"]])
PY
done
