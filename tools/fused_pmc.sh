#!/bin/bash
# SQ counters of the fused rollout kernel (200 steps per launch): tools/fused_pmc.sh <tag under gpurun_out> <variant|default> <mode>
set -u
TAG=$1; VAR=$2; MODE=$3
R=${GRAFT_REPO_ROOT:-$PWD}
if [ "$VAR" != default ]; then export AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_$VAR.so; fi
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS"
cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  eval "C=\$P$pass"
  out=$R/gpurun_out/$TAG/${VAR}_${MODE}/p$pass; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -- python3 $R/tools/fused_pmc_child.py $MODE > $out/out.log 2> $out/err.log || { echo "pass $pass FAILED"; tail -3 $out/err.log; }
done
echo "== $VAR $MODE (per launch of 200 steps)"
python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG/${VAR}_${MODE} rollout_kernel
