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
# slots 0 and 7 are s_memrealtime (100 MHz wall clock, common to all XCDs); 1..6 are s_memtime (shader clock, per XCD)
span = (s[:, 7].max() - s[:, 0].min()) * 10e-3
print("worlds %d, wavefronts %d" % (n, waves))
print("first wave start -> last wave end (wall): %.2f us; per-wave lifetime: median %.2f us, p90 %.2f, max %.2f" %
      (span, np.median(s[:, 7] - s[:, 0]) * 1e-2, np.percentile(s[:, 7] - s[:, 0], 90) * 1e-2, (s[:, 7] - s[:, 0]).max() * 1e-2))
print("start skew (first -> last wave start): %.2f us" % ((s[:, 0].max() - s[:, 0].min()) * 1e-2))
blk = s.reshape(-1, 16, 8)
print("per-block lifetime (first start -> last end): median %.2f us  p90 %.2f  max %.2f" %
      (np.median(blk[:, :, 7].max(1) - blk[:, :, 0].min(1)) * 1e-2, np.percentile(blk[:, :, 7].max(1) - blk[:, :, 0].min(1), 90) * 1e-2,
       (blk[:, :, 7].max(1) - blk[:, :, 0].min(1)).max() * 1e-2))
names2 = ["philox->fast path", "fast->exact", "exact->outputs+list", "list->barrier", "barrier->group reseed"]
for i in range(1, 6):
    d = s[:, i + 1] - s[:, i]
    print("%-28s median %7.0f  p90 %7.0f  p99 %7.0f  max %7.0f  (shader cycles)" % (names2[i - 1], np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max()))
cyc = (s[:, 6] - s[:, 1])
wall = (s[:, 7] - s[:, 0]) * 10.0
print("shader clock estimate: %.2f GHz (cycles 1->6 over wall 0->7, lower bound)" % np.median(cyc / wall))
