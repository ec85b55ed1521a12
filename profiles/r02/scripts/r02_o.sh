#!/bin/bash
# runtime knobs: where the kernel arguments live, how graphs are replayed
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
{
for r in 1 2; do
echo "== defaults";                        python tools/ab.py --rounds 1 default@2 default@0 2>&1 | grep us/step
echo "== HIP_FORCE_DEV_KERNARG=1";         HIP_FORCE_DEV_KERNARG=1 python tools/ab.py --rounds 1 default@2 default@0 2>&1 | grep us/step
echo "== HIP_FORCE_DEV_KERNARG=0";         HIP_FORCE_DEV_KERNARG=0 python tools/ab.py --rounds 1 default@2 default@0 2>&1 | grep us/step
echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python tools/ab.py --rounds 1 default@2 default@0 2>&1 | grep us/step
echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/ab.py --rounds 1 default@2 default@0 2>&1 | grep us/step
echo "== DEBUG_HIP_KERNARG_COPY_OPT=0";    DEBUG_HIP_KERNARG_COPY_OPT=0 python tools/ab.py --rounds 1 default@2 2>&1 | grep us/step
echo "== ROC_USE_FGS_KERNARG=0";           ROC_USE_FGS_KERNARG=0 python tools/ab.py --rounds 1 default@2 2>&1 | grep us/step
done
echo "== bench driver flags, defaults"; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['launch_us'], r['regions_ms'])"
echo "== bench driver flags, ROC_ACTIVE_WAIT_TIMEOUT=100000"; ROC_ACTIVE_WAIT_TIMEOUT=100000 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['launch_us'], r['regions_ms'])"
} > $O/knobs.txt 2>&1
cat $O/knobs.txt
