#!/usr/bin/env python3
"""Interleaved A/B timing of library variants on ONE GPU box (each measurement in a fresh process).
usage: python tools/ab.py [--rounds 3] [--envs N] name1 name2 ...   (name = variant under lib/variants or 'default', name@mode = restart mode 0 | 1 | 2)"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--tables", action="store_true", help="one obstacle list per world (BENCH8, each obstacle moved by up to 3 units)")
ap.add_argument("names", nargs="+")
args = ap.parse_args()

CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import torch
from aquaticgymenv_amd import presets, _capi
from aquaticgymenv_amd.batched import BatchedAqua
n, steps = %d, %d
obst = presets.BENCH8
if os.environ.get("AQUA_AB_TABLES") == "1":
    import numpy as np
    rng = np.random.RandomState(7)
    obst = np.repeat(presets.BENCH8[None], n, axis=0).astype(np.float64)
    obst[:, :, 0:2] += rng.uniform(-3, 3, (n, 8, 2))
env = BatchedAqua(n, obstacles=obst, seed=0, auto_reset=int(os.environ.get("AQUA_RESET_MODE", "1")), device="cuda:0")
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 3, (100, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
graph = env.capture_rollout(100, actions=acts, keep_all=False)
for _ in range(3): graph.launch()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps // 100): graph.launch()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / (steps // 100 * 100))
print("%%.3f" %% best)
''' % (ROOT, args.envs, args.steps)

res = {n: [] for n in args.names}
for r in range(args.rounds):
    for name in args.names:
        env = dict(os.environ)
        name, _, mode = name.partition("@")
        env["AQUA_RESET_MODE"] = mode or "1"
        env["AQUA_AB_TABLES"] = "1" if args.tables else "0"
        if name != "default":
            env["AQUA_HIP_LIB"] = os.path.join(ROOT, "aquaticgymenv_amd", "lib", "variants", "libaqua_hip_%s.so" % name)
        else:
            env.pop("AQUA_HIP_LIB", None)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        try:
            res[name + ("@" + mode if mode else "")].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print("FAILED", name, out.stderr[-500:])
for name in args.names:
    v = res[name]
    if v:
        print("%-16s us/step: min %.2f  median %.2f  all %s" % (name, min(v), sorted(v)[len(v) // 2], " ".join("%.2f" % x for x in v)))
