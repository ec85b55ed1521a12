#!/usr/bin/env python3
"""us per step of the fused rollout at 262 144 worlds, 8 obstacles (shared table), per restart mode, for the shipped library
and for library variants (aquaticgymenv_amd/build.py build_variant), each in a fresh process, interleaved rounds.
usage: python tools/fused_time.py [--rounds R] default nomail ...      (names of lib/variants/libaqua_hip_<name>.so)
AQUA_FUSED_TIME_STORED_ONLY=1: stored uint8 actions only (variants built with -DAQUA_DEV_U8_ONLY have no other kind)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua
n, T = 262144, 500
out = []
for mode in ("next_step", "same_step", False):
    for actions in (("stored",) if __import__("os").environ.get("AQUA_FUSED_TIME_STORED_ONLY") else ("stored", "sampled")):
        env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=mode, device="cuda:0")
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.randint(0, 3, (T, env.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8) if actions == "stored" else None
        for _ in range(2):
            env.rollout(T, actions=acts, fused=True, keep_all=False)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.rollout(T, actions=acts, fused=True, keep_all=False); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / T)
        out.append("%%s/%%s %%.3f" %% (mode or "no_restart", actions, best))
        del env
print("FUSED " + "  ".join(out))
'''
names = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
names = [a for a in names if not a.isdigit()] or ["default"]
for rnd in range(rounds):
    for name in names:
        env = dict(os.environ)
        if name != "default":
            env["AQUA_HIP_LIB"] = os.path.join(ROOT, "aquaticgymenv_amd", "lib", "variants", "libaqua_hip_%s.so" % name)
        out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("FUSED")]
        print("round %d %-10s %s" % (rnd, name, lines[0][6:] if lines else "FAILED " + out.stderr[-600:]), flush=True)
