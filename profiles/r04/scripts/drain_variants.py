#!/usr/bin/env python3
"""Wall clock of a 20-step region (one graph replay between two stream events) by the way the host waits for it:
torch.cuda.synchronize() | stream.synchronize() | event.synchronize() | spinning on event.query() and then synchronize().
usage: python tools/r04/drain_variants.py [steps=20] [regions=200]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
regions = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 262144
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 3, (steps, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    graph = env.capture_rollout(steps, actions=acts, keep_all=False)
    for _ in range(5):
        graph.launch()
    torch.cuda.synchronize()

    def wait_device(e1):
        torch.cuda.synchronize()

    def wait_stream(e1):
        stream.synchronize()
        torch.cuda.synchronize()

    def wait_event(e1):
        e1.synchronize()
        torch.cuda.synchronize()

    def wait_spin(e1):
        while not e1.query():
            pass
        torch.cuda.synchronize()

    variants = [("torch.cuda.synchronize", wait_device), ("stream.synchronize", wait_stream), ("event.synchronize", wait_event),
                ("spin on event.query", wait_spin)]
    res = {name: ([], []) for name, _ in variants}
    for r in range(regions):
        for name, wait in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            graph.launch()
            e1.record(stream)
            wait(e1)
            res[name][0].append((time.perf_counter() - t0) * 1e6)
            res[name][1].append(e0.elapsed_time(e1) * 1e3)
    for name, _ in variants:
        w, e = res[name]
        print("%-24s wall per region: median %.1f us  p10 %.1f  p90 %.1f   events: median %.1f us   (%d steps: %.3f / %.3f us per step)" % (
            name, statistics.median(w), sorted(w)[len(w) // 10], sorted(w)[9 * len(w) // 10], statistics.median(e), steps,
            statistics.median(w) / steps, statistics.median(e) / steps))
