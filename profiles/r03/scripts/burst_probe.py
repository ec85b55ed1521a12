#!/usr/bin/env python3
"""What does a 20-step region cost beyond 20 x the steady-state launch?  One 20-step graph of 262 144 worlds, timed with HIP
events on its launch stream (e0 ahead of the launch, e1 behind it, then synchronize), under controlled conditions:
host idle time before the region, host time between e0 and the launch, a busy GPU ahead of e0 (a gate kernel)."""
import ctypes, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
env = BatchedAqua(262144, obstacles=presets.BENCH8, seed=0, auto_reset="next_step", device=dev)
env.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = torch.randint(0, 3, (100, env.ld), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
hist = torch.zeros((T, env.ld // 64), dtype=torch.int64, device=dev)
graph = env.capture_rollout(T, actions=acts, keep_all=False, done_history=hist)
g100 = env.capture_rollout(100, actions=acts, keep_all=False)
pad = torch.empty(1 << 24, dtype=torch.float32, device=dev)     # 64 MB fill: a ~25 us gate kernel


def spin(us):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e6 < us:
        pass


def region(idle_us=0, between_us=0, gate=False, ahead=0):
    torch.cuda.synchronize()
    spin(idle_us)
    for _ in range(ahead):
        g100.launch()
    if gate:
        pad.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    spin(between_us)
    graph.launch()
    e1.record(st)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    return e0.elapsed_time(e1) * 1e3 / T, wall / T


def show(label, **kw):
    for _ in range(3):
        region(**kw)
    v = [region(**kw) for _ in range(9)]
    ev = sorted(x[0] for x in v)
    wl = sorted(x[1] for x in v)
    print("%-66s events us/step: median %.3f min %.3f max %.3f | wall us/step median %.2f" % (label, ev[4], ev[0], ev[-1], wl[4]), flush=True)


for _ in range(3):
    g100.launch()
show("back to back (synchronize, then the region)")
show("200 us of host idle before the region", idle_us=200)
show("2 ms of host idle before the region", idle_us=2000)
show("20 us of host work between e0 and the launch", between_us=20)
show("100 us of host work between e0 and the launch", between_us=100)
show("a 64 MB fill queued ahead of e0 (GPU busy while the region is queued)", gate=True)
show("200 us idle, then the fill ahead of e0", idle_us=200, gate=True)
show("one 100-step replay queued ahead of e0", ahead=1)
show("200 us idle, one 100-step replay queued ahead of e0", idle_us=200, ahead=1)
