#!/bin/bash
O=gpurun_out/r03_k20_eager.txt
: > $O
show() { python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); ro = r['roofline']
        print('value %.4g  ms/step %.6f  launch_us %.3f  frac %.4f  regions %s  launch=%s' % (r['value'], r['ms_per_step'], ro['launch_us'], ro['frac'], ['%.3f' % v for v in ro['launch_us_regions']], r['config']['launch']))
"; }
for i in 1 2; do
echo "== k20 graph" >> $O; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show >> $O
echo "== k20 eager" >> $O; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --eager 2>/dev/null | show >> $O
done
echo "== k100 graph" >> $O; python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | show >> $O
echo "== k100 eager" >> $O; python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --eager 2>/dev/null | show >> $O
echo "== defaults eager" >> $O; python3 bench.py --no-cpu-baseline --eager 2>/dev/null | show >> $O
cat $O
