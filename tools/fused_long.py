#!/usr/bin/env python3
"""fused rollout at several lengths: does a longer launch run at a higher clock?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for mode in (2, 1):
    env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=mode, device="cuda:0")
    env.reset()
    env.rollout(200, fused=True, keep_all=False)
    torch.cuda.synchronize()
    for T in (100, 1000, 10000, 40000):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(T, fused=True, keep_all=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("mode %d fused T=%6d: %8.2f ms  %6.2f us/step  %6.1f G steps/s" % (mode, T, ms, ms * 1e3 / T, n * T / ms / 1e6), flush=True)
