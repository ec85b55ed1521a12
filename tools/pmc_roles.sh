#!/bin/bash
# VALU/SALU instruction counts of the next-step kernel by role: whole kernel, stepping blocks alone (nw), re-seeding blocks alone (nm)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
for v in default nw nm; do
  out=$R/gpurun_out/pmcroles/$v; mkdir -p $out
  if [ $v != default ]; then export AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_$v.so; else unset AQUA_HIP_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -- python3 $R/bench.py --steps 300 --warmup 200 --no-cpu-baseline > $out/bench.json 2> $out/err.log || echo FAILED $v
  echo "== $v"; python3 $R/tools/pmc_summary.py $out step_
done
