#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
python tools/ab.py --rounds 3 default@2 r1@2 ilnever@2 default@1 r1@1 default@0 r1@0 > $O/ab_regress.txt 2>&1
cat $O/ab_regress.txt
for n in 8388608 6291456 2097152; do echo "N=$n"; python tools/ab.py --rounds 2 --envs $n --steps 200 il0@2 ilnever@2 2>&1 | grep us/step; done > $O/ab_interleave2.txt 2>&1
cat $O/ab_interleave2.txt
