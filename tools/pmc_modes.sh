#!/bin/bash
# SQ counter breakdown of the step kernel for the three restart modes (two passes of 8 counters each)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"
for cfg in "--no-auto-reset" "--reset-mode 1" "--reset-mode 2"; do
  tag=$(echo "m$cfg" | tr -d " -")
  for pass in 1 2; do
    eval "C=\$P$pass"
    out=$R/gpurun_out/pmc3/$tag/p$pass; mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -- python3 $R/bench.py --steps 200 --warmup 100 --no-cpu-baseline $cfg > $out/bench.json 2> $out/err.log || echo FAILED
  done
  echo "== $cfg"; python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3/$tag step_
done
