#!/bin/bash
# same-step restart of large batches: step_kernel (cur) against step_tile_kernel (sstile: from 2 M worlds on), then the bit-identity check
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/contact6; mkdir -p $O
AQUA_CHECK_MODE=same_step python3 tools/r04/compare_builds.py sstile cur 2097152 30 > $O/sst_check.txt 2>&1; cat $O/sst_check.txt
for n in 2097152 4194304 8388608 16777216; do
  python3 tools/ab.py --rounds 3 --envs $n --steps 200 cur@1 sstile@1 > $O/ab_sst_$n.txt 2>&1; echo "== $n"; cat $O/ab_sst_$n.txt
done
