#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/contact4; mkdir -p $O
python3 tools/ab.py --rounds 3 st0@2 st9@2 st11@2 st0pw@2 st9pw@2 > $O/ab_stagger4.txt 2>&1; cat $O/ab_stagger4.txt
AQUA_CHECK_MODE=same_step python3 tools/r04/compare_builds.py sst st0 > $O/sst_check.txt 2>&1; cat $O/sst_check.txt
python3 tools/ab.py --rounds 3 st0@1 sst@1 > $O/ab_sst.txt 2>&1; cat $O/ab_sst.txt
python3 tools/ab.py --rounds 2 --envs 4194304 --steps 200 st0@1 sst@1 > $O/ab_sst_4m.txt 2>&1; cat $O/ab_sst_4m.txt
python3 tools/ab.py --rounds 2 --envs 16777216 --steps 200 st0@1 sst@1 > $O/ab_sst_16m.txt 2>&1; cat $O/ab_sst_16m.txt
python3 tools/r04/fused_ab.py duty0 duty1 duty0 duty1 > $O/fused_duty.txt 2>&1; cat $O/fused_duty.txt
for n in 524288 2097152 8388608 16777216; do
  python3 tools/ab.py --rounds 2 --envs $n --steps 200 st0@2 nst2@2 > $O/ab_nst2_$n.txt 2>&1; echo "== $n"; cat $O/ab_nst2_$n.txt
done
