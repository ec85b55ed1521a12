#!/usr/bin/env python3
"""A 20-step region timed three ways on one box: the graph between two recorded stream events (what bench.py did), eager
launches between the same two stream events, and eager launches with the events ATTACHED to the first and the last launch
(rollout(events=LaunchEvents())).  Also a 2 000-step region the same ways."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua, LaunchEvents

n = 262144
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
acts = torch.randint(0, 3, (100, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
g20 = env.capture_rollout(20, actions=acts, keep_all=False)
g100 = env.capture_rollout(100, actions=acts, keep_all=False)
for _ in range(5):
    g100.launch()
torch.cuda.synchronize()
lev = LaunchEvents()


def region(kind, steps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    if kind == "graph, stream events":
        e0.record()
        for _ in range(max(1, steps // 100)):
            (g20 if steps == 20 else g100).launch()
        e1.record()
    elif kind == "eager, stream events":
        e0.record()
        for _ in range(max(1, steps // 100)):
            env.rollout(min(steps, 100), actions=acts, keep_all=False)
        e1.record()
    else:
        blocks = max(1, steps // 100)
        for b in range(blocks):
            ev = (lev.start if b == 0 else None, lev.stop if b == blocks - 1 else None)
            env.rollout(min(steps, 100), actions=acts, keep_all=False, events=ev)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - h0) * 1e6
    ms = lev.elapsed_ms() if kind.startswith("eager, launch") else e0.elapsed_time(e1)
    return ms * 1e3 / steps, wall / steps


for steps in (20, 2000):
    for kind in ("graph, stream events", "eager, stream events", "eager, launch events", "graph, stream events", "eager, launch events"):
        for _ in range(3):
            region(kind, steps)
        r = [region(kind, steps) for _ in range(9)]
        ev = sorted(x[0] for x in r)
        wl = sorted(x[1] for x in r)
        print("%5d steps, %-22s: events us/step median %.3f min %.3f max %.3f | wall us/step median %.2f" %
              (steps, kind, ev[4], ev[0], ev[-1], wl[4]), flush=True)
