#!/bin/bash
# what the done-mask exchange costs the step stream with ONE rank (the copy to the own receive buffer; RCCL's all-gather)
O=gpurun_out/r03_exchange_overhead_one_rank.txt
: > $O
show() { python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); ro = r['roofline']; c = r['config']
        print('value %.4g  ms/step %.6f  launch_us %.3f  regions_ms %s  exchange=%s/%s' % (r['value'], r['ms_per_step'], ro['launch_us'], ['%.3f' % v for v in r['regions_ms']], c.get('done_mask_exchange_kind'), c.get('done_mask_copy_engine')))
"; }
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533"
for K in "" "--steps 20 --warmup 5"; do
  echo "=== bench.py --no-cpu-baseline $K" >> $O
  echo "-- plain" >> $O; python3 bench.py --no-cpu-baseline $K 2>/dev/null | show >> $O
  echo "-- one rank under torch.distributed.run, no exchange" >> $O; $TR bench.py --gpus 1 --no-cpu-baseline $K 2>/dev/null | show >> $O
  echo "-- + exchange ipc, copy by wavefronts" >> $O; $TR bench.py --gpus 1 --no-cpu-baseline --force-exchange --exchange ipc $K 2>/dev/null | show >> $O
  echo "-- + exchange ipc, copy by hipMemcpyAsync" >> $O; $TR bench.py --gpus 1 --no-cpu-baseline --force-exchange --exchange ipc --copy-engine dma $K 2>/dev/null | show >> $O
  echo "-- + exchange rccl" >> $O; $TR bench.py --gpus 1 --no-cpu-baseline --force-exchange --exchange rccl $K 2>/dev/null | show >> $O
done
cat $O
