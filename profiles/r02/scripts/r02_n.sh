#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
python tools/ab.py --tables --rounds 3 default@1 vialds@1 default@0 vialds@0 > $O/ab_tables_lds.txt 2>&1
cat $O/ab_tables_lds.txt
bash tools/pmc_big.sh 16777216 > $O/pmc_n16m_interleaved.txt 2>&1
mv $R/gpurun_out/pmcbig $O/pmcbig_interleaved
AQUA_HIP_LIB=$R/aquaticgymenv_amd/lib/variants/libaqua_hip_ilnever.so bash tools/pmc_big.sh 16777216 > $O/pmc_n16m_head_of_grid.txt 2>&1
mv $R/gpurun_out/pmcbig $O/pmcbig_head_of_grid
tail -20 $O/pmc_n16m_interleaved.txt; tail -20 $O/pmc_n16m_head_of_grid.txt
bash tools/profile_r02.sh r02 > $O/profile_r02.log 2>&1 || { tail -30 $O/profile_r02.log; exit 1; }
tail -60 $O/profile_r02.log
