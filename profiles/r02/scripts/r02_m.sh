#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
python tools/ab.py --tables --rounds 2 default@2 nw@2 nm@2 default@0 > $O/ab_tables_roles.txt 2>&1
cat $O/ab_tables_roles.txt
