#!/usr/bin/env python3
"""Why does a 20-step graph that starts on an idle GPU run slower per step than the same graph in a stream of replays?
In-graph event time of ONE replay of a T-step graph (262 144 worlds, BENCH8, next-step restart) under different
conditions before it.  usage: python tools/burst.py [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
env = BatchedAqua(262144, obstacles=presets.BENCH8, seed=0, auto_reset="next_step", device="cuda:0")
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 3, (100, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
graph = env.capture_rollout(T, actions=acts, keep_all=False, timing=True)
a = torch.randn(2048, 2048, device="cuda")
pad = torch.empty(1 << 22, device="cuda")


def filler(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        torch.mm(a, a)


def one(label, before):
    vals = []
    for _ in range(7):
        torch.cuda.synchronize()
        before()
        graph.launch()
        torch.cuda.synchronize()
        vals.append(graph.elapsed_ms() * 1e3 / T)
    vals.sort()
    print("%-58s us/step median %.3f  min %.3f  max %.3f" % (label, vals[3], vals[0], vals[-1]))


for _ in range(3):
    graph.launch()
one("idle GPU, then the graph", lambda: None)
one("host sleeps 2 ms, then the graph", lambda: time.sleep(0.002))
one("5 ms of matmuls queued, graph queued behind them", lambda: filler(5))
one("5 ms of matmuls, synchronise, then the graph", lambda: (filler(5), torch.cuda.synchronize()))
one("one 16 MB fill queued, graph queued behind it", lambda: pad.fill_(1.0))
one("3 untimed replays queued, graph queued behind them", lambda: [graph.launch() for _ in range(3)])
one("50 untimed replays queued, graph queued behind them", lambda: [graph.launch() for _ in range(50)])
