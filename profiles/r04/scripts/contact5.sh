#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/contact5; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit $rc; }
python3 tools/ab.py --rounds 3 nolate@2 cur@2 nolate@1 cur@1 nolate@0 cur@0 > $O/ab_late_loads_262144.txt 2>&1; cat $O/ab_late_loads_262144.txt
for n in 65536 131072 524288 1048576; do
  python3 tools/ab.py --rounds 2 --envs $n --steps 500 nolate@2 cur@2 lateil@2 nolate@1 cur@1 sstile@1 nolate@0 cur@0 > $O/ab_$n.txt 2>&1; echo "== $n"; cat $O/ab_$n.txt
done
for n in 2097152 16777216; do
  python3 tools/ab.py --rounds 2 --envs $n --steps 200 cur@2 lateil@2 cur@1 sstile@1 nolate@0 cur@0 > $O/ab_$n.txt 2>&1; echo "== $n"; cat $O/ab_$n.txt
done
AQUA_HIP_LIB= python3 tools/r04/fused_ab.py duty0 cur duty0 cur > $O/fused_duty.txt 2>&1; cat $O/fused_duty.txt
bash profiles/r04/scripts/tables_coop.sh
