#!/bin/bash
O=gpurun_out/r03_k20_settle.txt
: > $O
show() { python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); ro = r['roofline']
        print('value %.4g  ms/step %.6f  launch_us %.3f  frac %.4f  regions %s' % (r['value'], r['ms_per_step'], ro['launch_us'], ro['frac'], ['%.3f' % v for v in ro['launch_us_regions']]))
"; }
for i in 1 2 3; do
for s in 0 200 1000; do
echo "== k20 --settle-us $s" >> $O; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --settle-us $s 2>/dev/null | show >> $O
done; done
cat $O
