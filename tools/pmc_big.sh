#!/bin/bash
# SQ counters of the step kernel at a streaming size (default 16.7 M worlds): is the launch VALU-issue bound?
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-16777216}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_BUSY_CU_CYCLES"
for pass in 1 2; do
  eval "C=\$P$pass"
  out=$R/gpurun_out/pmcbig/n$N/p$pass; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -- python3 $R/bench.py --envs $N --steps 100 --warmup 100 --no-cpu-baseline > $out/bench.json 2> $out/err.log || echo FAILED pass $pass
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcbig/n$N step_
