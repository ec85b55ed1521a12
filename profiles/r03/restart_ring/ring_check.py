#!/usr/bin/env python3
"""Round 3: first check + timing of auto_reset 3 (restart rings) against auto_reset 2 on the GPU.
 (1) worlds that have not restarted yet evolve bit-identically in both modes; a restart reports reward 0 / term 0, its
     state is a valid placement, the ring head advances by the number of restarts;
 (2) us per step of a 100-step graph, both modes, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

dev = torch.device("cuda:0")
args = [v for v in sys.argv[1:] if not v.startswith("--")]
n = int(args[0]) if args else 262144
time_only = "--time-only" in sys.argv
envs = {m: BatchedAqua(n, obstacles=presets.BENCH8, seed=3, auto_reset=m, device=dev) for m in ("next_step", "next_step_ring")}
for e in envs.values():
    e.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = torch.randint(0, 3, (100, envs["next_step"].ld), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
a, b = envs["next_step"], envs["next_step_ring"]
assert torch.equal(a.state, b.state)
fresh = torch.ones(n, dtype=torch.bool, device=dev)          # worlds that have not restarted in either run
restarts = 0
for t in range(0 if time_only else 60):
    if t > 0:                                                 # worlds that restart in this step leave the comparison
        fresh = fresh & ~(b.time[:n] == -1 - ((t - 1) & 1)) & ~(a.time[:n] == -1 - ((t - 1) & 1))
    a.step(acts[t]); b.step(acts[t])
    torch.cuda.synchronize()
    same = fresh
    assert torch.equal(a.term[:n][same], b.term[:n][same]), "term differs at step %d" % t
    assert torch.equal(a.reward[:n][same], b.reward[:n][same]), "reward differs at step %d" % t
    assert torch.equal(a.state[:, :n][:, same], b.state[:, :n][:, same]), "state differs at step %d" % t
    assert torch.equal(a.time[:n][same], b.time[:n][same]), "time differs at step %d" % t
    # worlds of b restarted in this step: marker restart_code(t), reward 0, term 0, a placement inside the border band
    rc = -3 - (t & 1)
    r = b.time[:n] == rc
    if r.any():
        assert (b.term[:n][r] == 0).all() and (b.reward[:n][r] == 0).all()
        st = b.state[:, :n][:, r]
        assert (st[0] >= 2.5).all() and (st[0] <= 97.5).all() and (st[3] >= 2.5).all() and (st[4] <= 97.5).all()
        assert ((st[0] - st[3]) ** 2 + (st[1] - st[4]) ** 2 > 25.0).all()
    restarts += int(r.sum())
heads = b.ring_head[:, :, 0].max(dim=0).values.to(torch.int64)
if not time_only:
  print("60 steps: %d restarts; ring heads sum %d (groups %d); still-fresh worlds %d" % (restarts, int(heads.sum()), heads.numel(), int(fresh.sum())))
if not time_only:
  assert int(heads.sum()) in (restarts, restarts + int((b.time[:n] == -1 - (59 & 1)).sum())), "heads do not match the restarts"
# (2) timing
graphs = {m: e.capture_rollout(100, actions=acts, keep_all=False) for m, e in envs.items()}
for gr in graphs.values():
    for _ in range(3):
        gr.launch()
torch.cuda.synchronize()
for rep in range(3):
    for m, gr in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.launch()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)
        print("%-16s %.3f us per step  (frac %.3f)" % (m, us, 62 * n / us / 1e3 / 8000), flush=True)
