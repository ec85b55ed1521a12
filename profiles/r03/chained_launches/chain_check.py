#!/usr/bin/env python3
"""chained launches (rollout(chained=True)) against the serialised launches: same results, and the time per step.
usage: python tools/r03/chain_check.py [--time-only]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aquaticgymenv_amd import presets                    # noqa: E402
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402


def same(n, steps):
    g = torch.Generator(device="cuda").manual_seed(n)
    envs = [BatchedAqua(n, obstacles=presets.BENCH8, seed=5, auto_reset=2, device="cuda:0") for _ in range(2)]
    acts = torch.randint(0, 3, (steps, envs[0].ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
    out = []
    for env, chained in zip(envs, (False, True)):
        env.reset()
        r, t = env.rollout(steps, actions=acts, done_history=True if False else None, chained=chained)
        r2, t2 = env.rollout(steps, actions=acts, chained=chained)        # a second chain behind the first
        torch.cuda.synchronize()
        out.append((env.state.clone(), env.time.clone(), r, t, r2, t2))
    ok = all(torch.equal(a[..., :n] if a.dim() > 1 else a[:n], b[..., :n] if b.dim() > 1 else b[:n]) for a, b in zip(*out))
    print("N %7d, 2 x %d steps: chained == serialised: %s, error word %d, episodes ended %d" %
          (n, steps, ok, envs[1].chain_errors(), int((out[0][3][:, :n] != 0).sum())), flush=True)
    return ok and envs[1].chain_errors() == 0


def timing(n=262144, block=100, blocks=20):
    env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.randint(0, 3, (block, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
    graph = env.capture_rollout(block, actions=acts, keep_all=False)
    for name in ("graph", "chained", "eager", "graph", "chained"):
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            e0.record()
            for _ in range(blocks):
                if name == "graph":
                    graph.launch()
                else:
                    env.rollout(block, actions=acts, keep_all=False, chained=name == "chained")
            e1.record()
            h1 = time.perf_counter()
            torch.cuda.synchronize()
            print("%-8s %d x %d steps: %.3f us/step by events, host enqueue %.2f us/step" %
                  (name, blocks, block, e0.elapsed_time(e1) * 1e3 / (block * blocks), (h1 - h0) * 1e6 / (block * blocks)), flush=True)
    print("error word", env.chain_errors())


if __name__ == "__main__":
    ok = True
    if "--time-only" not in sys.argv:
        for n, steps in ((1000, 250), (4113, 250), (70001, 250), (262144, 250)):
            ok = same(n, steps) and ok
    if ok:
        timing()
    sys.exit(0 if ok else 1)
