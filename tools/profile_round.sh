#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:  tools/profile_round.sh <tag> [bench args...]
#   gpurun_out/<tag>/stats   : --kernel-trace --stats            (kernel durations)
#   gpurun_out/<tag>/fetch   : --kernel-trace --pmc FETCH_SIZE   (separate pass, MI355X_MICROARCH.md "HBM")
#   gpurun_out/<tag>/write   : --kernel-trace --pmc WRITE_SIZE   (separate pass)
# rocprofv3 gets the program itself after "--" (python3 bench.py ...), never a wrapper.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/stats_bench.json 2> $OUT/stats_err.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/fetch_bench.json 2> $OUT/fetch_err.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/write_bench.json 2> $OUT/write_err.log || exit 1
python3 $R/tools/profile_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
