#!/usr/bin/env python3
"""us per step of the fused rollouts (shared table; per-world tables of 8 rows) for library variants, fresh process each.
usage: fused_ab.py name1 name2 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua
n, T = 262144, 500
out = []
for tables in (False, True):
    for mode in ("next_step", "same_step") + ((False,) if __import__("os").environ.get("AQUA_FUSED_AB_NO_RESTART") else ()):
        obst = presets.BENCH8
        if tables:
            rng = np.random.RandomState(7)
            obst = np.repeat(presets.BENCH8[None], n, axis=0).astype(np.float64)
            obst[:, :, 0:2] += rng.uniform(-3, 3, (n, 8, 2))
        env = BatchedAqua(n, obstacles=obst, seed=0, auto_reset=mode, device="cuda:0")
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.randint(0, 3, (T, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
        env.rollout(T, actions=acts, fused=True, keep_all=False)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.rollout(T, actions=acts, fused=True, keep_all=False); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / T)
        out.append("%%s/%%s %%.3f" %% ("tables" if tables else "shared", mode or "no_restart", best))
        del env
print("FUSED " + "  ".join(out))
'''
for name in sys.argv[1:]:
    env = dict(os.environ)
    if name != "default":
        env["AQUA_HIP_LIB"] = os.path.join(ROOT, "aquaticgymenv_amd", "lib", "variants", "libaqua_hip_%s.so" % name)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("FUSED")]
    print("%-10s %s" % (name, lines[0][6:] if lines else "FAILED " + out.stderr[-400:]))
