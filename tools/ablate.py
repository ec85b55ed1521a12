#!/usr/bin/env python3
"""Timing ablations of the step kernel on one GPU (graph replay, HIP events).  Not part of the product.
usage: python tools/ablate.py [--envs N] [--steps K]"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd import _capi, presets
from aquaticgymenv_amd.batched import BatchedAqua

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--vecs", default="1,2,4")
args = ap.parse_args()
CH = 100


def time_variant(vec, auto_reset, obst, mode, fused=False):
    _capi.lib.aqua_set_vector_width(vec)
    env = BatchedAqua(args.envs, obstacles=obst, seed=0, auto_reset=auto_reset, device="cuda:0")
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    actions = None if mode == "sample" else torch.randint(0, 3, (CH, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
    graph = env.capture_rollout(CH, actions=actions, keep_all=False, fused=fused)
    for _ in range(2):
        graph.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps // CH):
        graph.launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (args.steps // CH * CH)
    return us


print("envs", args.envs)
for vec in [int(v) for v in args.vecs.split(",")]:
    for auto_reset, obst_name, mode in itertools.product((True, False), ("bench8", "none"), ("u8", "sample")):
        obst = presets.BENCH8 if obst_name == "bench8" else presets.NONE
        us = time_variant(vec, auto_reset, obst, mode)
        print("vec %d auto_reset %-5s obst %-6s actions %-6s : %7.2f us/step  %6.1f G steps/s  %5.1f%% of 8TB/s" %
              (vec, auto_reset, obst_name, mode, us, args.envs / us / 1e3, 62 * args.envs / us / 1e3 / 8000 * 100), flush=True)
us = time_variant(1, True, presets.BENCH8, "sample", fused=True)
print("fused rollout (graph of 1 launch x %d steps): %.2f us/step" % (CH, us))
