#!/bin/bash
# Round 5's evidence on ONE box, part B (no tracer): the bench lines of every configuration, the one-rank exchange lines with
# their extra legs (extras.exchange_ab), the N > 1 path rehearsed with two and four ranks on the one GPU, the fused rollouts
# and the long per-world tables.      tools/bench_lines_r05.sh <tag under gpurun_out>      (needs profiles/traffic.json of this build)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
T=${1:-r05final}
O=$R/gpurun_out/$T
mkdir -p $O/rehearsal
b() { name=$1; shift; python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail $O/bench_$name.err; exit 1; }; }
b driver --gpus 1 --steps 20 --warmup 5
b driver_2 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b driver_3 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
b default --extras --per-world-tables --no-cpu-baseline
b n16m --envs 16777216 --steps 100 --warmup 20 --no-cpu-baseline
b continuous --continuous --no-cpu-baseline
b same_step --reset-mode 1 --no-cpu-baseline
b no_restart --no-auto-reset --no-cpu-baseline
b n4096_noobst --envs 4096 --no-obstacles --no-cpu-baseline
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port"
$TR 29541 bench.py --gpus 1 --no-cpu-baseline --force-exchange > $O/rehearsal/one_rank_ipc_main.json 2> $O/rehearsal/one_rank_ipc_main.err
$TR 29542 bench.py --gpus 1 --no-cpu-baseline --force-exchange --exchange rccl > $O/rehearsal/one_rank_rccl_main.json 2> $O/rehearsal/one_rank_rccl_main.err
$TR 29543 bench.py --gpus 1 --no-cpu-baseline --force-exchange --steps 20 --warmup 5 > $O/rehearsal/one_rank_ipc_main_k20.json 2> $O/rehearsal/one_rank_ipc_main_k20.err
python3 bench.py --gpus 2 --ranks-on-one-gpu --envs 131072 --steps 20 --warmup 5 --no-cpu-baseline > $O/rehearsal/two_ranks_one_gpu.json 2> $O/rehearsal/two_ranks_one_gpu.err
python3 bench.py --gpus 4 --ranks-on-one-gpu --envs 65536 --steps 20 --warmup 5 --no-cpu-baseline > $O/rehearsal/four_ranks_one_gpu.json 2> $O/rehearsal/four_ranks_one_gpu.err
AQUA_TEST_EXCHANGE_FAIL=ab-stall python3 bench.py --gpus 2 --ranks-on-one-gpu --envs 131072 --steps 20 --warmup 5 --no-cpu-baseline --ab-deadline 12 > $O/rehearsal/two_ranks_leg_stalls.json 2> $O/rehearsal/two_ranks_leg_stalls.err
python3 tools/fused_time.py --rounds 2 default > $O/fused_time.txt 2>&1
python3 tools/tables_long_time.py 9 17 32 64 --modes next_step,same_step,none,none_fresh --reps 2 > $O/tables_long_time.txt 2>&1
for f in $O/bench_*.json $O/rehearsal/*.json; do python3 - $f <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[1].split("/")[-1], "UNREADABLE", e); sys.exit(0)
c=r["config"]; ro=r["roofline"]
ab=(r.get("extras") or {}).get("exchange_ab")
print(sys.argv[1].split("/")[-1], "value %.4g by_events %.4g  us/step wall %.3f events %.3f  frac %.4f live %.4f wall %.4f kernel_only %s  first5 %.4f  outliers %s  n_gpus %d exchange %s  beyond_warmup %s" % (
    r["value"], r["value_by_events"], r["ms_per_step"]*1e3, ro["launch_us"], ro["frac"], ro["frac_live"], ro["frac_by_wall"], ro["kernel_only"].get("frac"),
    ro["first_regions"]["frac"], r["regions_outliers"]["by_events"]["n"], r["n_gpus"], c.get("done_mask_exchange_kind"), c.get("untimed_steps_beyond_warmup")))
if ab:
    print("    exchange_ab:", {k: ({kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("wall_us_per_step", "event_us_per_step", "frac", "own_block_intact", "error", "regions")} if isinstance(v, dict) else v) for k, v in ab.items() if k != "what"})
PY
done
cat $O/fused_time.txt $O/tables_long_time.txt
