#!/usr/bin/env python3
"""Experiment: S independent sub-batches advanced concurrently on S streams with EAGER launches (one host thread
per stream, each in the C-side launch loop of aqua_rollout_f32).  Do kernels of different streams overlap?
usage: python tools/multistream_eager.py [--envs N] [--streams 1,2,4] [--mode 2]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--streams", default="1,2,4")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--mode", type=int, default=2)
args = ap.parse_args()
CH = 100
for S in [int(v) for v in args.streams.split(",")]:
    n = args.envs // S
    envs, acts, streams = [], [], []
    for s in range(S):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            e = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, env_offset=s * n, auto_reset=args.mode, device="cuda:0")
            e.reset()
            g = torch.Generator(device="cuda").manual_seed(s)
            a = torch.randint(0, 3, (CH, e.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
            e.rollout(CH, actions=a, keep_all=False)
        envs.append(e); acts.append(a); streams.append(st)
    torch.cuda.synchronize()

    def worker(s):
        with torch.cuda.stream(streams[s]):
            for _ in range(args.reps):
                envs[s].rollout(CH, actions=acts[s], keep_all=False)

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(s,)) for s in range(S)]
    for t in threads: t.start()
    for t in threads: t.join()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    us = t1 * 1e6 / (args.reps * CH)
    print("streams %d x %7d worlds: %6.2f us per full-batch step (host enqueue done after %.0f%% of the time)  %6.1f G steps/s  %5.1f%% of 8 TB/s" %
          (S, n, us, 100 * t_host / t1, args.envs / us / 1e3, 62 * args.envs / us / 1e3 / 8000 * 100), flush=True)
