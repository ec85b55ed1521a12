#!/usr/bin/env python3
"""three fused rollouts of 200 steps at 262 144 worlds (for rocprofv3 --pmc; tools/fused_pmc.sh).  argv: restart mode"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua
mode = {"none": False}.get(sys.argv[1], sys.argv[1])
n, T = 262144, 200
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=mode, device="cuda:0")
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 3, (T, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
for _ in range(3):
    env.rollout(T, actions=acts, fused=True, keep_all=False)
torch.cuda.synchronize()
