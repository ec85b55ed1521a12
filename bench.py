#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched AquaEnv step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE batched step (one launch of the fused step kernel) over one batch of worlds.
Workload at N=1 (BASELINE.json configs[2], the configuration the metric is quoted on):
262 144 worlds, discrete uint8 actions (pre-generated, i.i.d. uniform), 4 circle + 4 rectangle
obstacles, waves on, Philox noise, auto-reset on.  N > 1: the same batch PER GPU (weak scaling),
range-partitioned global world indices, done-mask all-gather over RCCL on a side stream.

Steps are queued as replays of a captured HIP graph of CHUNK steps (+ a remainder of eager
launches so that exactly K steps are timed).  Inputs are resident in HBM before the timed region.
One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

A_DISCRETE = 62      # algorithmic bytes per world-step, SURVEY.md 8(d): 33 read + 29 written
A_CONTINUOUS = 69
HBM_PEAK_GBPS = 8000.0
CHUNK = 100          # steps per captured HIP graph
GATHER_EVERY = 5     # graphs per done-mask all-gather (N > 1): one [500][words] block per exchange


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=262144, help="worlds per GPU")
    ap.add_argument("--continuous", action="store_true", help="configs[3]: continuous actions")
    ap.add_argument("--no-obstacles", action="store_true", help="configs[1]-style: no obstacles")
    ap.add_argument("--no-auto-reset", action="store_true", help="diagnostic only: finished worlds are not restarted")
    ap.add_argument("--reset-mode", type=int, default=2, choices=(1, 2),
                    help="restart of finished worlds: 2 = during the next step (default, fastest), 1 = inside the same launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--eager", action="store_true", help="no HIP graph: one C-side launch loop per chunk")
    ap.add_argument("--vec", type=int, default=0, help="worlds per lane (0 = auto)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the done-mask exchange (side stream, double buffer) even on one GPU: rehearsal of the N > 1 path")
    ap.add_argument("--extras", action="store_true", help="also time the fused rollout kernel and a 16M-world point")
    ap.add_argument("--per-world-tables", action="store_true",
                    help="separate line (extras.per_world_tables): every world has its own 8-obstacle list")
    return ap.parse_args()


def run_steps(env, graphs, actions, n_steps, exchange, hist, state):
    """exactly n_steps batched steps.  Full chunks replay captured graphs; every GATHER_EVERY chunks the
    [GATHER_EVERY * CHUNK][words] block of done masks they wrote is all-gathered on the side stream
    (double-buffered: the step stream only waits for the gather that last read the buffer it is about to
    overwrite, i.e. the one queued two blocks ago)."""
    full, rem = divmod(n_steps, CHUNK)
    for _ in range(full):
        c = state["chunk"]
        buf, w = (c // GATHER_EVERY) & 1, c % GATHER_EVERY
        if exchange is not None and w == 0:
            exchange.wait_source(buf)
        if graphs is not None:
            graphs[buf][w].launch()
        else:
            env.rollout(CHUNK, actions=actions, keep_all=False, done_history=hist[buf][w * CHUNK:(w + 1) * CHUNK])
        if exchange is not None and w == GATHER_EVERY - 1:
            exchange.gather_async(hist[buf], source_id=buf)
        state["chunk"] = c + 1
    if rem:
        env.rollout(rem, actions=actions, keep_all=False)


def cpu_baseline(args, obstacles):
    """the CPU path timed beside the GPU number, on this host's cores (rank 0, N=1 only):
    'port'   = the reference-style one-world-per-object numpy port (oracle.ScalarPort), 1 core;
    'port_c' = the float64 C oracle driving float32 state, OpenMP over all cores."""
    import numpy as np
    from oracle.aqua_oracle import COracle, time_scalar_port
    steps, secs = time_scalar_port(obstacles, args.continuous, budget_s=args.cpu_seconds)
    base = {"value": steps / secs, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d random-action steps of ONE world (reset on done), oracle.ScalarPort, %.1f s" % (steps, secs)}
    orc = COracle()
    n = 262144
    st = np.zeros((7, n), dtype=np.float32)
    tt = np.zeros(n, dtype=np.int32)
    orc.reset(st, tt, obstacles=obstacles, waves=1, seed=0, tick=0)
    orc.rollout_f32(st, tt, 2, obstacles=obstacles, continuous=args.continuous, seed=0, tick0=1)
    t0 = time.perf_counter()
    T, done = 0, 0.0
    while done < min(args.cpu_seconds, 8.0):
        orc.rollout_f32(st, tt, 10, obstacles=obstacles, continuous=args.continuous, seed=0, tick0=3 + T)
        T += 10
        done = time.perf_counter() - t0
    c = {"value": n * T / done, "unit": "env-steps/s", "cores": orc.threads(), "kind": "port",
         "sample": "%d steps of %d worlds, C float64 oracle (oracle/aqua_oracle.c) with OpenMP, %.1f s" % (T, n, done)}
    return base, c


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from aquaticgymenv_amd import _capi, presets
    from aquaticgymenv_amd.batched import BatchedAqua
    from aquaticgymenv_amd.sharded import DoneMaskExchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    if args.vec:
        _capi.lib.aqua_set_vector_width(args.vec)

    n = args.envs
    obstacles = presets.NONE if args.no_obstacles else presets.BENCH8
    env = BatchedAqua(n, obstacles=obstacles, continuous=args.continuous, seed=0, env_offset=rank * n,
                      auto_reset=0 if args.no_auto_reset else args.reset_mode, device=dev)
    env.reset()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    if args.continuous:
        actions = torch.rand((CHUNK, 2, env.ld), device=dev, generator=gen) * 0.3 + 0.2
    else:
        actions = torch.randint(0, 3, (CHUNK, env.ld), device=dev, generator=gen, dtype=torch.int64).to(torch.uint8)
    words = env.ld // 64
    hist = [torch.zeros((GATHER_EVERY * CHUNK, words), dtype=torch.int64, device=dev) for _ in range(2)]
    exchange = DoneMaskExchange(GATHER_EVERY * CHUNK, words, dev) if (world > 1 or args.force_exchange) else None
    graph = None
    if not args.eager:
        graph = [[env.capture_rollout(CHUNK, actions=actions, keep_all=False,
                                      done_history=hist[b][w * CHUNK:(w + 1) * CHUNK]) for w in range(GATHER_EVERY)]
                 for b in range(2)]
    progress = {"chunk": 0}

    run_steps(env, graph, actions, args.warmup, exchange, hist, progress)
    if exchange is not None:
        exchange.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run_steps(env, graph, actions, args.steps, exchange, hist, progress)
    e1.record()
    if exchange is not None:
        exchange.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())

    # sanity on the timed work: worlds did move and episodes did end (no cached / skipped work)
    ended = int((hist[0][:CHUNK].cpu().numpy().view(np.uint64) != 0).sum())
    assert env._tick >= args.steps + args.warmup and (ended > 0 or args.no_auto_reset)

    result = None
    if rank == 0:
        a_bytes = A_CONTINUOUS if args.continuous else A_DISCRETE
        steps_per_s = world * n * args.steps / wall
        # the step kernel is the only kernel in the timed stream: its average launch period on the
        # launch stream (HIP events, boundary gaps included -> a lower bound on the kernel's own rate)
        launch_s = ev_ms * 1e-3 / args.steps
        achieved = a_bytes * n / launch_s / 1e9
        result = {
            "metric": "env-steps/sec at batch=262144; achieved HBM GB/s vs roofline; 1/2/4/8-GPU scaling",
            "value": steps_per_s, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch %d worlds/GPU, %s actions, %s, waves on, auto-reset (%s), HIP graph of %d steps"
                       % (n, "continuous f32x2" if args.continuous else "discrete u8",
                          "no obstacles" if args.no_obstacles else "4 circle + 4 rect obstacles",
                          "off" if args.no_auto_reset else ("next-step" if args.reset_mode == 2 else "same-step"), CHUNK),
                       "baseline_config": "configs[3]" if args.continuous else ("configs[1]-like" if args.no_obstacles else "configs[2]"),
                       "worlds_per_gpu": n, "global_worlds": world * n, "parallelism": "range-partition x%d" % world,
                       "launch": "eager" if args.eager else "hipGraph"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "algorithmic_bytes_per_world_step": a_bytes, "launch_us": launch_s * 1e6,
                         "note": "launch_us = HIP-event time of the timed region / launches (includes the "
                                 "inter-kernel boundary); kernel-only duration: profiles/"},
        }
        result["roofline"]["traffic"] = committed_traffic(n, args)
        if world == 1 and not args.no_cpu_baseline:
            base, c = cpu_baseline(args, obstacles)
            result["cpu_baseline"] = base
            result["cpu_baseline_c"] = c
        if args.extras and world == 1:
            result["extras"] = extras(env, torch, n, a_bytes)
        if args.per_world_tables and world == 1:
            result.setdefault("extras", {})["per_world_tables"] = per_world_tables(torch, np, presets, BatchedAqua, n, dev)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def committed_traffic(n, args):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate passes, KB -> bytes, FETCH_SIZE doubled: on gfx950 it reports half of a
    coalesced stream, MI355X_MICROARCH.md "HBM").  null when no profile matches this configuration."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None
    key = "n%d_%s_%s" % (n, "cont" if args.continuous else "disc", "k0" if args.no_obstacles else "k8")
    row = table.get(key)
    if not row:
        return None
    return 2.0 * row["FETCH_SIZE_per_launch_raw"] * 1024.0 + row["WRITE_SIZE_per_launch_raw"] * 1024.0


def per_world_tables(torch, np, presets, BatchedAqua, n, dev):
    """separate line: every world with its own obstacle list (BENCH8, each obstacle moved by up to +-3 units per
    world), same-step restart inside the step launch, HIP graph of 100 steps.  Algorithmic bytes per
    world-step: 62 + 24 per obstacle row read (6 float32 per row)."""
    rng = np.random.RandomState(7)
    tables = np.repeat(presets.BENCH8[None], n, axis=0).astype(np.float64)
    tables[:, :, 0:2] += rng.uniform(-3, 3, (n, 8, 2))
    env = BatchedAqua(n, obstacles=tables, seed=0, auto_reset="same_step", device=dev)
    env.reset()
    g = torch.Generator(device=dev).manual_seed(99)
    actions = torch.randint(0, 3, (CHUNK, env.ld), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
    graph = env.capture_steps_per_world(CHUNK, actions)
    for _ in range(3):
        graph.launch()
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * CHUNK)
    a_bytes = A_DISCRETE + 24 * 8
    return {"env_steps_per_s": n / us * 1e6, "us_per_step": us, "launches_per_step": 1,
            "algorithmic_bytes_per_world_step": a_bytes, "achieved_GBps": a_bytes * n / us / 1e3,
            "frac_of_8TBps": a_bytes * n / us / 1e3 / HBM_PEAK_GBPS,
            "note": "per-world obstacle tables [K][6][N] float32, finished worlds restarted inside the step launch (same-step)"}


def extras(env, torch, n, a_bytes):
    """separate lines that must not be mixed into `value`: the fused multi-step kernel (state in
    registers: 5 B per world-step: reward 4 + term 1, actions sampled on the device)."""
    out = {}
    T = 2000
    env.rollout(T, fused=True, keep_all=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.rollout(T, fused=True, keep_all=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    out["fused_rollout"] = {"env_steps_per_s": n * T / (ms * 1e-3), "steps": T, "bytes_per_world_step": 5,
                            "us_per_step": ms * 1e3 / T, "actions": "sampled on device",
                            "note": "ONE launch for all steps, state in registers: the issue-bound floor of this arithmetic"}
    return out


if __name__ == "__main__":
    main()
