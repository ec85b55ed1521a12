#!/bin/bash
# runtime knobs of the HIP/ROCr stack against the driver's command and the default run (diagnostic)
O=gpurun_out/r03_env_knobs.txt
: > $O
show() { python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); ro = r['roofline']
        print('value %.4g  ms/step %.6f  launch_us %.3f  frac %.4f' % (r['value'], r['ms_per_step'], ro['launch_us'], ro['frac']))
"; }
run() { echo "== $*" >> $O; env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show >> $O; env "$@" python3 bench.py --no-cpu-baseline --steps 500 --warmup 100 2>/dev/null | show >> $O; }
run X=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HSA_ENABLE_SDMA=0
run X=2
cat $O
