#!/usr/bin/env python3
"""Experiment (round 3, VERDICT lever b): the batch as S independent sub-batches whose step launches overlap.
Worlds are independent (aqua.py:135-213 has no cross-world term), so the S chains of step launches have no edges
between them; a chain's inter-kernel boundary (drain + dispatch, ~1.5 us) can hide behind the other chains' work.

Forms timed, per FULL-batch step:
  one      the shipped path: one env, one graph of T steps
  streams  S envs (env_offset = s n / S), one graph each, launched back to back on S streams
  forked   ONE graph that holds the S chains (fork at its head, join at its tail), one hipGraphLaunch
each as (a) steady state: 10 replays of T = 100 steps, and (b) the driver's shape: ONE T = 20 graph from an idle GPU.
usage: python tools/r03/split_streams.py [--envs N] [--splits 1,2,4]"""
import argparse
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aquaticgymenv_amd import presets, _capi
from aquaticgymenv_amd.batched import BatchedAqua

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--splits", default="1,2,4")
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _capi.lib


def make_envs(S):
    n = args.envs // S
    envs = []
    for s in range(S):
        e = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, env_offset=s * n, auto_reset="next_step", device=dev)
        e.reset()
        g = torch.Generator(device=dev).manual_seed(s)
        e._acts = torch.randint(0, 3, (100, e.ld), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
        envs.append(e)
    return envs


def queue_rollout(e, T, stream):
    e._sync_device_tick()
    rc = lib.aqua_rollout_f32(ctypes.byref(e.params), e._blob_ptr(), e.K, e.num_envs, e.env_offset, e.state.data_ptr(), e.ld,
                              e.time.data_ptr(), T, e._acts.data_ptr(), _capi.ACT_U8, 0, e._acts.stride(0), e.seed, 0,
                              e._tick_dev.data_ptr(), e.reward.data_ptr(), e.term.data_ptr(), 0, e.done_bits.data_ptr(), 0,
                              None, 2, 1, ctypes.c_void_p(stream.cuda_stream))
    _capi.check(rc, "rollout")


def capture_forked(envs, T, streams):
    """one graph: chain s on streams[s], forked from / joined into streams[0]"""
    torch.cuda.synchronize()
    handle = ctypes.c_void_p()
    s0 = streams[0]
    _capi.check(lib.aqua_graph_begin(ctypes.c_void_p(s0.cuda_stream)), "begin")
    try:
        fork = torch.cuda.Event()
        fork.record(s0)
        for st in streams[1:]:
            st.wait_event(fork)
        for e, st in zip(envs, streams):
            queue_rollout(e, T, st)
        for st in streams[1:]:
            j = torch.cuda.Event()
            j.record(st)
            s0.wait_event(j)
    finally:
        _capi.check(lib.aqua_graph_end(ctypes.c_void_p(s0.cuda_stream), ctypes.byref(handle)), "end")
    return handle


def capture_single(e, T, stream):
    torch.cuda.synchronize()
    handle = ctypes.c_void_p()
    _capi.check(lib.aqua_graph_begin(ctypes.c_void_p(stream.cuda_stream)), "begin")
    try:
        queue_rollout(e, T, stream)
    finally:
        _capi.check(lib.aqua_graph_end(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(handle)), "end")
    return handle


def time_jobs(jobs, T, replays, main):
    """jobs: [(graph handle, stream)]; returns us per full-batch step between an event ahead of all and one behind all"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _, st in jobs:
        if st is not main:
            st.wait_event(e0)
    for _ in range(replays):
        for h, st in jobs:
            _capi.check(lib.aqua_graph_launch(h, ctypes.c_void_p(st.cuda_stream)), "launch")
    for _, st in jobs:
        if st is not main:
            main.wait_stream(st)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * T)


for S in [int(v) for v in args.splits.split(",")]:
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    envs = make_envs(S)
    for T, replays, reps, label in ((100, 10, 3, "steady T=100 x10"), (20, 1, 9, "burst  T=20  x1 ")):
        sep = [(capture_single(e, T, st), st) for e, st in zip(envs, streams)]
        forms = {"streams": sep}
        if S > 1:
            forms["forked"] = [(capture_forked(envs, T, streams), streams[0])]
        for name, jobs in forms.items():
            for _ in range(2):
                time_jobs(jobs, T, replays, streams[0])
            vals = [time_jobs(jobs, T, replays, streams[0]) for _ in range(reps)]
            us = statistics.median(vals)
            print("S=%d %-8s %s: %6.3f us per full-batch step (min %.3f max %.3f)  frac %.3f" %
                  (S, name if S > 1 else "one", label, us, min(vals), max(vals), 62 * args.envs / us / 1e3 / 8000), flush=True)
    del envs
