#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/contact9; mkdir -p $O
python3 tools/ab.py --rounds 3 --tables cur@2 tlate@2 cur@1 tlate@1 cur@0 tlate@0 > $O/ab_tables_late.txt 2>&1; cat $O/ab_tables_late.txt
for K in 16 32; do for v in cur tlate; do echo -n "$v  "; AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_$v.so python3 tools/r03/tables_step_time.py $K 2>/dev/null | tail -1; done; done | tee $O/tables_late_long.txt
