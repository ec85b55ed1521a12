#!/bin/bash
# rocprofv3 evidence of round 3, all on ONE box: the driver's command, the default bench line, the streaming size,
# continuous actions (kernel-trace stats + FETCH_SIZE / WRITE_SIZE in separate passes); then bench lines of the same box
# without the tracer, the one-rank exchange lines, and the N > 1 path rehearsed with two ranks on the one GPU
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
T=${1:-r03}
bash tools/profile_round.sh $T/k20 --steps 20 --warmup 5 || exit 1
bash tools/profile_round.sh $T/n262144 || exit 1
bash tools/profile_round.sh $T/n16m --envs 16777216 --steps 100 --warmup 20 || exit 1
bash tools/profile_round.sh $T/n262144_cont --continuous --steps 500 --warmup 100 || exit 1
O=$R/gpurun_out/$T
python3 tools/make_traffic.py $O n262144_disc_k8=n262144 n16777216_disc_k8=n16m n262144_cont_k8=n262144_cont > $O/traffic.log 2>&1 || { cat $O/traffic.log; exit 1; }
cp profiles/traffic.json $O/traffic.json
b() { name=$1; shift; python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail $O/bench_$name.err; exit 1; }; }
b driver --gpus 1 --steps 20 --warmup 5
b default --extras --per-world-tables
b n16m --envs 16777216 --steps 100 --warmup 20 --no-cpu-baseline
b continuous --continuous --no-cpu-baseline
b same_step --reset-mode 1 --no-cpu-baseline
b no_restart --no-auto-reset --no-cpu-baseline
b n4096_noobst --envs 4096 --no-obstacles --no-cpu-baseline
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541"
$TR bench.py --gpus 1 --no-cpu-baseline --force-exchange > $O/bench_one_rank_exchange_ipc.json 2>/dev/null
$TR bench.py --gpus 1 --no-cpu-baseline --force-exchange --exchange rccl > $O/bench_one_rank_exchange_rccl.json 2>/dev/null
$TR bench.py --gpus 1 --no-cpu-baseline --force-exchange --steps 20 --warmup 5 > $O/bench_one_rank_exchange_ipc_k20.json 2>/dev/null
python3 bench.py --gpus 2 --ranks-on-one-gpu --envs 131072 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_two_ranks_one_gpu_rehearsal.json 2>/dev/null
for f in $O/bench_*.json; do python3 - $f <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "UNREADABLE", e); sys.exit(0)
c=r["config"]
print(sys.argv[1].split("/")[-1], "value %.4g  ms/step %.6f  launch_us %.3f  frac %.4f  traffic %s  n_gpus %d ranks_seen %s exchange %s" % (r["value"], r["ms_per_step"], r["roofline"]["launch_us"], r["roofline"]["frac"], r["roofline"]["traffic"], r["n_gpus"], c.get("ranks_seen"), c.get("done_mask_exchange_kind")))
PY
done
