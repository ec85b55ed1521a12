#!/bin/bash
# the N > 1 path of bench.py on ONE GPU with 2 / 4 ranks sharing cuda:0 (the box's process guard allows six processes on the GPU) (gloo rendezvous, IPC exchange between the
# processes), clean and with injected faults: every run must end with a line; timings of several processes on one GPU mean nothing
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/${1:-r04/rehearsal}; mkdir -p $O
run() { name=$1; faults=$2; shift 2
  AQUA_TEST_EXCHANGE_FAIL=$faults timeout -k 10 400 python3 bench.py --ranks-on-one-gpu --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $? <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); c = r["config"]
    print("%-28s rc %s  ranks %d  exchange %s  own block intact %s  peers' episodes %s  note %s" % (
        sys.argv[1].split("/")[-1], sys.argv[2], c["ranks_seen"], c["done_mask_exchange_kind"],
        (r["sanity"]["done_mask_exchange_last_block"] or {}).get("own_block_intact"),
        (r["sanity"]["done_mask_exchange_last_block"] or {}).get("episodes_in_peer_blocks"), (c["done_mask_exchange_note"] or "")[:110]))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "rc", sys.argv[2], "NO LINE", e)
PY
}
run ranks2_k20 "" --gpus 2 --envs 131072 --steps 20 --warmup 5
run ranks4_k20 "" --gpus 4 --envs 65536 --steps 20 --warmup 5
run ranks4_600 "" --gpus 4 --envs 65536 --steps 600 --warmup 100
run ranks4_open open --gpus 4 --envs 65536 --steps 20 --warmup 5
run ranks4_probe probe --gpus 4 --envs 65536 --steps 20 --warmup 5
run ranks4_stall stall --gpus 4 --envs 65536 --steps 20 --warmup 5 --soft-deadline 8
run ranks4_hard_stall hard-stall --gpus 4 --envs 65536 --steps 20 --warmup 5 --setup-deadline 10 --rendezvous-timeout 20
