#!/bin/bash
# Round 5's evidence on ONE box, part A (rocprofv3): the driver's command, the default bench line, the streaming size and
# continuous actions (kernel-trace stats + FETCH_SIZE / WRITE_SIZE in separate passes), the SQ counters of the benchmarked
# kernel; profiles/traffic.json of THIS build.     tools/profile_r05.sh <tag under gpurun_out, e.g. r05final>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
T=${1:-r05final}
bash tools/profile_round.sh $T/k20 --steps 20 --warmup 5 || exit 1
bash tools/profile_round.sh $T/n262144 || exit 1
bash tools/profile_round.sh $T/n16m --envs 16777216 --steps 100 --warmup 20 || exit 1
bash tools/profile_round.sh $T/n262144_cont --continuous --steps 500 --warmup 100 || exit 1
bash tools/pmc_sq.sh $T/pmc_sq > $R/gpurun_out/$T/pmc_sq.txt 2>&1
O=$R/gpurun_out/$T
python3 tools/make_traffic.py $O n262144_disc_k8=n262144 n262144_disc_k8_steps20=k20 n16777216_disc_k8=n16m n262144_cont_k8=n262144_cont > $O/traffic.log 2>&1 || { cat $O/traffic.log; exit 1; }
cp profiles/traffic.json $O/traffic.json
tail -5 $O/pmc_sq.txt
