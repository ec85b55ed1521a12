#!/usr/bin/env python3
"""How much of a 20-step region's event time is host time between the first recorded event and the graph's launch?
The same graph between the same two stream events, regions back to back: (a) RolloutGraph.launch() as bench.py calls it,
(b) the C entry point called directly with everything marshalled beforehand, (c) as (b) with the events recorded by the
C ABI's own wrappers where it has them.  usage: python tools/r04/launch_gap.py [steps=20] [regions=400]"""
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from aquaticgymenv_amd import presets, _capi
from aquaticgymenv_amd.batched import BatchedAqua

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
regions = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env = BatchedAqua(262144, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 3, (steps, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    graph = env.capture_rollout(steps, actions=acts, keep_all=False)
    for _ in range(50):
        graph.launch()
    torch.cuda.synchronize()
    launch_c, handle, st = _capi.lib.aqua_graph_launch, graph._handle, ctypes.c_void_p(stream.cuda_stream)

    def as_bench(e0, e1):
        e0.record(stream)
        graph.launch()
        e1.record(stream)

    def direct(e0, e1):
        e0.record(stream)
        launch_c(handle, st)
        e1.record(stream)
        env._tick += steps
        env._device_tick += steps

    variants = [("RolloutGraph.launch()", as_bench), ("C entry point, marshalled", direct)]
    res = {name: ([], []) for name, _ in variants}
    for r in range(regions):
        for name, fn in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            fn(e0, e1)
            torch.cuda.synchronize()
            res[name][0].append((time.perf_counter() - t0) * 1e6)
            res[name][1].append(e0.elapsed_time(e1) * 1e3)
    for name, _ in variants:
        w, e = res[name]
        print("%-28s events: median %.2f us per region = %.3f per step   wall: median %.1f us = %.3f per step" % (
            name, statistics.median(e), statistics.median(e) / steps, statistics.median(w), statistics.median(w) / steps))
