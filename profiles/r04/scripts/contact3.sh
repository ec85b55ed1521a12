#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/contact3; mkdir -p $O
python3 tools/ab.py --rounds 3 st0@2 st2@2 st8@2 st9@2 st10@2 st2kp@2 st0kp@2 > $O/ab_stagger3.txt 2>&1; cat $O/ab_stagger3.txt
python3 tools/r04/compare_builds.py nst st0 > $O/nst_check.txt 2>&1; cat $O/nst_check.txt
for n in 524288 1048576 2097152 4194304 8388608 16777216; do
  python3 tools/ab.py --rounds 2 --envs $n --steps 200 st0@2 nst@2 nstwb@2 nstwt@2 > $O/ab_nst_$n.txt 2>&1; echo "== $n"; cat $O/ab_nst_$n.txt
done
