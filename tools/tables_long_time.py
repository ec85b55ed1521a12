#!/usr/bin/env python3
"""per-world obstacle tables of K rows, one launch per step (replayed 50-step graph): us per step by restart mode.
usage: python tools/tables_long_time.py K [K ...] [--modes next_step,same_step,none] [--reps R]
Under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass) the per-kernel rows give the traffic per launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402


def tables_for(n, K, rng):
    """K rows per world, sized so that about a quarter of a world is blocked (boat radius included)"""
    t = np.zeros((n, K, 5))
    t[:, :, 0:2] = rng.uniform(10, 90, (n, K, 2))
    kind = rng.randint(0, 2, (n, K)).astype(np.float64)
    scale = max(0.05, ((0.28 * 1.0e4 / (np.pi * K)) ** 0.5 - 2.5) / 6.0)
    t[:, :, 2] = kind
    t[:, :, 3] = np.where(kind == 0, rng.uniform(2, 10, (n, K)), rng.uniform(5, 15, (n, K))) * scale
    t[:, :, 4] = np.where(kind == 0, 0.0, rng.uniform(5, 15, (n, K)) * scale)
    return t


def main():
    args = [a for a in sys.argv[1:]]
    modes = ["next_step", "same_step", "none"]
    reps = 3
    if "--modes" in args:
        modes = args[args.index("--modes") + 1].split(",")
        del args[args.index("--modes"):args.index("--modes") + 2]
    if "--reps" in args:
        reps = int(args[args.index("--reps") + 1])
        del args[args.index("--reps"):args.index("--reps") + 2]
    n, steps = 262144, 50
    for K in [int(a) for a in args] or [32, 64]:
        tables = tables_for(n, K, np.random.RandomState(7))
        acts = torch.randint(0, 3, (steps, n), dtype=torch.uint8, device="cuda:0")
        row = []
        for mode in modes:
            # "none_fresh": no restarts, but every timed launch starts from a fresh reset() -- the worlds are still inside the
            # 100 x 100 world, among their obstacles (in "none" they have long left it by the time the clock starts, and a world
            # far from every obstacle never takes the second look or the float64 path)
            fresh = mode == "none_fresh"
            env = BatchedAqua(n, obstacles=tables, device="cuda:0", seed=3, auto_reset=False if mode.startswith("none") else mode)
            env.reset()
            g = env.capture_rollout(steps, actions=acts)
            for _ in range(2):
                g.launch()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(reps):
                if fresh:
                    env.reset()
                    torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(1 if fresh else 4):
                    g.launch()
                b.record()
                torch.cuda.synchronize()
                best = min(best, 1e3 * a.elapsed_time(b) / ((1 if fresh else 4) * steps))
            row.append("%s %.2f" % (mode, best))
            del g, env
        print("K=%d  %s   (algorithmic %d B per world-step)" % (K, "  ".join(row), 62 + 24 * K), flush=True)


if __name__ == "__main__":
    main()
