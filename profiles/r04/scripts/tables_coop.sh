#!/bin/bash
# per-world tables of 9-64 rows, one launch per step: the re-seeding groups' rows fetched cooperatively (tcoop) against two at a
# time inside the attempt loop (tworld, round 3's form); then the parity tests of the long tables on the new form
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/tables; mkdir -p $O
for K in 9 17 32 64; do
  for v in tworld cur; do
    echo -n "$v  " >> $O/tables_coop.txt
    AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_$v.so python3 tools/r03/tables_step_time.py $K 2>/dev/null | tail -1 >> $O/tables_coop.txt
  done
done
cat $O/tables_coop.txt
