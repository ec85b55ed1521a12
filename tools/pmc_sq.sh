#!/bin/bash
# SQ counters of the benchmarked kernel (configs[2], next-step restart), three passes of <= 8 counters:
#   tools/pmc_sq.sh <tag under gpurun_out> [bench args]
# rocprofv3 gets the program itself after "--"; --pmc passes carry --kernel-trace only (no other trace domain)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_IFETCH SQ_INST_CYCLES_VMEM"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"
cd /tmp && export TMPDIR=/tmp
for pass in 1 2 3; do
  eval "C=\$P$pass"
  out=$R/gpurun_out/$TAG/p$pass; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -- python3 $R/bench.py --steps 200 --warmup 100 --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log || { echo "pass $pass FAILED"; tail -3 $out/err.log; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG step_
