#!/usr/bin/env python3
"""The in-tile next-step kernel (a library built with -DAQUA_NS_TILE_MIN=...) against the role-split one (another
library), bit for bit: the same batch stepped by each in a child process of its own, every buffer compared.
usage: [AQUA_CHECK_MODE=same_step] nst_check.py <variant A> <variant B> [worlds] [steps]      (variant = name under lib/variants, or 'default')"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, hashlib
sys.path.insert(0, %r)
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua
n, steps = %d, %d
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=5, auto_reset=__import__("os").environ.get("AQUA_CHECK_MODE", "next_step"), device="cuda:0", env_offset=1024)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 3, (steps, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
rew, term = env.rollout(steps, actions=acts, keep_all=True, done_history=True)
graph = env.capture_rollout(7, actions=acts, keep_all=False)
graph.launch(); graph.launch()
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (env.state, env.time, rew, term, env.reward, env.term, env.done_bits):
    h.update(t.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest(), int((term != 0).sum()))
'''


def run(name, n, steps):
    env = dict(os.environ)
    if name != "default":
        env["AQUA_HIP_LIB"] = os.path.join(ROOT, "aquaticgymenv_amd", "lib", "variants", "libaqua_hip_%s.so" % name)
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, n, steps)], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")]
    if not lines:
        sys.exit("%s failed:\n%s" % (name, out.stderr[-2000:]))
    return lines[0]


a, b = sys.argv[1], sys.argv[2]
sizes = [int(sys.argv[3])] if len(sys.argv) > 3 else [524288 + 77, 2097152, 1000]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
bad = 0
for n in sizes:
    da, db = run(a, n, steps), run(b, n, steps)
    same = da == db
    bad += not same
    print("%9d worlds x %d steps: %s %s   %s | %s" % (n, steps, a, b, "EQUAL" if same else "DIFFERENT", da.split()[2] + " episodes ended"))
    if not same:
        print("   ", da, "\n   ", db)
sys.exit(1 if bad else 0)
