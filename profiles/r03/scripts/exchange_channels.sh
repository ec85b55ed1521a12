#!/bin/bash
# one-rank rehearsal of the done-mask exchange with RCCL limited to 1 / 2 channels (cheap experiment)
set -e
out=gpurun_out/r03_exchange_channels.txt
: > $out
run() { echo "== $1" >> $out; shift; "$@" 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('value %.4g  ms/step %.6f  launch_us %.3f  regions_ms %s' % (r['value'], r['ms_per_step'], r['roofline']['launch_us'], ['%.3f' % v for v in r['regions_ms']]))
" >> $out; }
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
run plain python3 bench.py --no-cpu-baseline
run dist_exchange $TR bench.py --no-cpu-baseline --force-exchange
NCCL_MAX_NCHANNELS=1 run dist_exchange_1ch $TR bench.py --no-cpu-baseline --force-exchange
NCCL_MAX_NCHANNELS=2 run dist_exchange_2ch $TR bench.py --no-cpu-baseline --force-exchange
run k20 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5
run k20_dist_exchange $TR bench.py --no-cpu-baseline --force-exchange --steps 20 --warmup 5
NCCL_MAX_NCHANNELS=1 run k20_dist_exchange_1ch $TR bench.py --no-cpu-baseline --force-exchange --steps 20 --warmup 5
cat $out
