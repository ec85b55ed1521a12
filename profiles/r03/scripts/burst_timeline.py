#!/usr/bin/env python3
"""Where do the 12-16 us go that a 20-step region costs beyond 20 steady launches?  First wavefront start and last
wavefront end of EVERY launch of a 20-step graph (s_memrealtime, 10 ns; build -DAQUA_STAMPS=3: libaqua_hip_stamps3.so),
for a region behind a synchronize (what bench.py --steps 20 times) and for the same graph inside a stream of replays,
next to the HIP-event time of the region."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aquaticgymenv_amd import _capi, presets
from aquaticgymenv_amd.batched import BatchedAqua

n, K = 262144, 20
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
blocks = n // 256 + n // 1024
stamps = torch.zeros((32, blocks, 4, 2), dtype=torch.int64, device="cuda")
_capi.lib.aqua_debug_set_stamps.argtypes = [ctypes.c_void_p]
_capi.check(_capi.lib.aqua_debug_set_stamps(stamps.data_ptr()), "set stamps")
acts = torch.randint(0, 3, (100, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
g = env.capture_rollout(K, actions=acts, keep_all=False)
for _ in range(5):
    g.launch()
torch.cuda.synchronize()


def timeline(first_tick):
    s = stamps.cpu().numpy().astype(np.float64)
    rows = []
    for j in range(K):
        t = s[(first_tick + j) & 31]
        start = t[:, :, 0]
        end = t[:, :, 1]
        rows.append((start[start > 0].min(), end.max()))
    return rows


def show(name, rows, ev_us):
    s0 = rows[0][0]
    dur = [(e - s) * 1e-2 for s, e in rows]
    gap = [(rows[j + 1][0] - rows[j][1]) * 1e-2 for j in range(K - 1)]
    per = [(rows[j + 1][0] - rows[j][0]) * 1e-2 for j in range(K - 1)]
    span = (rows[-1][1] - s0) * 1e-2
    print("%s: events %.1f us (%.3f per step); first start -> last end %.1f us (%.3f per step); events - span = %.1f us" %
          (name, ev_us, ev_us / K, span, span / K, ev_us - span))
    print("   launch spans : " + " ".join("%.2f" % d for d in dur))
    print("   gaps         : " + " ".join("%.2f" % d for d in gap))
    print("   start-start  : " + " ".join("%.2f" % d for d in per), flush=True)


for rep in range(4):
    torch.cuda.synchronize()
    time.sleep(0.002)
    stamps.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tick = env._tick
    e0.record()
    g.launch()
    e1.record()
    torch.cuda.synchronize()
    show("region behind a synchronize", timeline(tick), e0.elapsed_time(e1) * 1e3)
for rep in range(2):
    torch.cuda.synchronize()
    stamps.zero_()
    for _ in range(10):
        g.launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tick = env._tick
    e0.record()
    g.launch()
    e1.record()
    for _ in range(3):
        g.launch()
    torch.cuda.synchronize()
    # the stamps of the last 32 launches survive: the timed graph is launches [-80, -60) -- gone; time the LAST graph instead
    show("last graph of a stream of replays (its own events are of the 4th from last)", timeline(env._tick - K), e0.elapsed_time(e1) * 1e3)
