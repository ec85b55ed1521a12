#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04/contact8; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "beyond_one_launch or clipped_action" > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
python3 tools/ab.py --rounds 2 --envs 16777216 --steps 200 cur@2 g4@2 g16@2 > $O/ab_group_16m.txt 2>&1; cat $O/ab_group_16m.txt
python3 tools/ab.py --rounds 2 --envs 2097152 --steps 200 cur@2 g4@2 > $O/ab_group_2m.txt 2>&1; cat $O/ab_group_2m.txt
python3 tools/ab.py --rounds 2 cur@2 g4@2 > $O/ab_group_262144.txt 2>&1; cat $O/ab_group_262144.txt
