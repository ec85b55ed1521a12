#!/bin/bash
# the driver's command, and the two clocks of a one-graph region side by side (stream events vs event-record nodes in the graph)
O=gpurun_out/r03_k20_event_methods.txt
: > $O
show() { python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); ro = r['roofline']
        print('value %.4g  ms/step %.6f  launch_us %.3f  frac %.4f  stream-event regions %s  graph-node regions %s' % (r['value'], r['ms_per_step'], ro['launch_us'], ro['frac'], ['%.3f' % v for v in ro['launch_us_regions']], ['%.3f' % v for v in (ro.get('launch_us_graph_nodes_regions') or [])]))
"; }
for i in 1 2 3; do
  echo "== python3 bench.py --gpus 1 --steps 20 --warmup 5 (run $i)" >> $O
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show >> $O
  echo "== the same + --graph-node-events (run $i)" >> $O
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --graph-node-events 2>/dev/null | show >> $O
done
echo "== defaults (2000 steps)" >> $O
python3 bench.py --no-cpu-baseline 2>/dev/null | show >> $O
echo "== one rank, --force-exchange: ipc / rccl, 2000-step and 20-step regions" >> $O
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531"
for x in ipc rccl; do
  echo "-- $x defaults" >> $O; $TR bench.py --no-cpu-baseline --force-exchange --exchange $x 2>/dev/null | show >> $O
  echo "-- $x k20" >> $O; $TR bench.py --no-cpu-baseline --force-exchange --exchange $x --steps 20 --warmup 5 2>/dev/null | show >> $O
done
echo "-- plain process --force-exchange (no process group) k20" >> $O; python3 bench.py --no-cpu-baseline --force-exchange --steps 20 --warmup 5 2>/dev/null | show >> $O
cat $O
