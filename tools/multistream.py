#!/usr/bin/env python3
"""Experiment: S independent sub-batches advanced concurrently on S streams (one HIP graph each).
usage: python tools/multistream.py [--envs N] [--streams 1,2,4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--streams", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
CH = 100
for S in [int(v) for v in args.streams.split(",")]:
    n = args.envs // S
    envs, graphs, streams = [], [], []
    for s in range(S):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            e = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, env_offset=s * n, auto_reset=True, device="cuda:0")
            e.reset()
            g = torch.Generator(device="cuda").manual_seed(s)
            acts = torch.randint(0, 3, (CH, e.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
            gr = e.capture_rollout(CH, actions=acts, keep_all=False)
            gr.launch()
        envs.append(e); graphs.append(gr); streams.append(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in streams:
        st.wait_event(e0)
    for _ in range(args.reps):
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                graphs[s].launch()
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (args.reps * CH)
    print("streams %d x %7d worlds: %6.2f us per full-batch step  %6.1f G steps/s  %5.1f%% of 8 TB/s" %
          (S, n, us, args.envs / us / 1e3, 62 * args.envs / us / 1e3 / 8000 * 100), flush=True)
