#!/usr/bin/env python3
"""Phase timing of the step kernel from s_memtime stamps (diagnostic build libaqua_hip_stamps.so).
AQUA_HIP_LIB=aquaticgymenv_amd/lib/variants/libaqua_hip_stamps.so python tools/stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aquaticgymenv_amd import _capi, presets
from aquaticgymenv_amd.batched import BatchedAqua

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=True, device="cuda:0")
env.reset()
waves = (n + 63) // 64
stamps = torch.zeros((waves, 8), dtype=torch.int64, device="cuda")
_capi.lib.aqua_debug_set_stamps.argtypes = [ctypes.c_void_p]
_capi.check(_capi.lib.aqua_debug_set_stamps(stamps.data_ptr()), "set stamps")
acts = torch.randint(0, 3, (64, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
env.rollout(60, actions=acts, keep_all=False)
names = ["start->loads issued+philox", "philox->fast path", "fast->exact", "exact->outputs+list", "list->barrier",
         "barrier->group reseed", "reseed->stores"]
rows = []
for rep in range(20):
    stamps.zero_()
    env.rollout(1, actions=acts, keep_all=False)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64)
    t0 = s[:, 0].min()
    rows.append((s, t0))
s, t0 = rows[-1]
print("worlds %d, wavefronts %d; s_memtime ticks (100 MHz constant clock? see below)" % (n, waves))
print("span of kernel (first stamp0 -> last stamp7): %.0f ticks" % (s[:, 7].max() - t0))
for i in range(8):
    rel = s[:, i] - t0
    print("stamp %d: min %8.0f  median %8.0f  p99 %8.0f  max %8.0f" % (i, rel.min(), np.median(rel), np.percentile(rel, 99), rel.max()))
for i in range(7):
    d = s[:, i + 1] - s[:, i]
    print("%-28s median %7.0f  p90 %7.0f  p99 %7.0f  max %7.0f" % (names[i], np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max()))
spans = [r[0][:, 7].max() - r[1] for r in rows]
print("kernel span over 20 launches: median %.0f min %.0f max %.0f ticks" % (np.median(spans), min(spans), max(spans)))
