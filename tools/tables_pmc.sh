#!/bin/bash
# Address-translation and L1 counters of the per-world table step kernels by restart mode (one kernel name per mode):
#   tools/tables_pmc.sh <tag under gpurun_out> <rows>
# rocprofv3 gets the program itself after "--"; --pmc passes carry --kernel-trace only
set -u
TAG=$1; K=${2:-64}
R=${GRAFT_REPO_ROOT:-$PWD}
P1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
P2="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
P3="GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp
for pass in 1 2 3; do
  eval "C=\$P$pass"
  out=$R/gpurun_out/$TAG/p$pass; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -- python3 $R/tools/tables_long_time.py $K --modes next_step,same_step,none --reps 1 > $out/out.log 2> $out/err.log || { echo "pass $pass FAILED"; tail -3 $out/err.log; }
done
for pat in "step_tables_kernel<0, 3, 0" "step_tables_kernel<0, 1, 0" "step_tables_kernel<0, 0, 0"; do echo "== $pat   (mode 3 = next-step in the tile, 1 = same-step, 0 = no restart)"; python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG "$pat"; done
