#!/usr/bin/env python3
"""per-world obstacle tables of K rows (sized so that about a quarter of a world is blocked), 262 144 worlds, one launch per
step as a replayed graph: us per step in the three restart modes.  usage: python tools/r03/tables_step_time.py K"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aquaticgymenv_amd.batched import BatchedAqua
n, K, steps = 262144, int(sys.argv[1]), 100
rng = np.random.RandomState(7)
tables = np.zeros((n, K, 5))
tables[:, :, 0:2] = rng.uniform(10, 90, (n, K, 2))
kind = rng.randint(0, 2, (n, K)).astype(np.float64)
scale = max(0.05, ((0.28 * 1.0e4 / (np.pi * K)) ** 0.5 - 2.5) / 6.0)
tables[:, :, 2] = kind
tables[:, :, 3] = np.where(kind == 0, rng.uniform(2, 10, (n, K)), rng.uniform(5, 15, (n, K))) * scale
tables[:, :, 4] = np.where(kind == 0, 0.0, rng.uniform(5, 15, (n, K)) * scale)
acts = torch.randint(0, 3, (steps, n), dtype=torch.uint8, device="cuda:0")
out = []
for mode in ("next_step", "same_step", False):
    env = BatchedAqua(n, obstacles=tables, device="cuda:0", seed=3, auto_reset=mode)
    env.reset()
    g = env.capture_rollout(steps, actions=acts, fused=False)
    for _ in range(3): g.launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): g.launch()
        b.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * a.elapsed_time(b) / (5 * steps))
    out.append("%s %.2f" % (mode, best))
print("K=%d  " % K + "  ".join(out))
