#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
python tools/ab.py --rounds 2 default@2 r1@2 nm@2 nm_r1@2 nw@2 p16s2@2 p16@2 prio0@2 > $O/ab_roles.txt 2>&1
cat $O/ab_roles.txt
