#!/usr/bin/env python3
"""Long soak of the fused rollout's restart hand-over (round 5: next-step = a re-seeding wavefront of its own and sequence
numbers in LDS, spin waits bounded by MAIL_SPIN_LIMIT): LAUNCHES fused launches of T steps each at 262 144 worlds against the
same steps by the per-step kernels (a replayed 100-step graph), stored actions; state, time markers and the last step's
outputs compared bit for bit after every launch.  usage: python tools/soak_fused.py [T=20000] [launches=5] [mode=next_step]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd import presets                    # noqa: E402
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    mode = sys.argv[3] if len(sys.argv) > 3 else "next_step"
    n, chunk = 262144, 100
    assert T % chunk == 0
    a = BatchedAqua(n, obstacles=presets.BENCH8, seed=5, auto_reset=mode, device="cuda:0")
    b = BatchedAqua(n, obstacles=presets.BENCH8, seed=5, auto_reset=mode, device="cuda:0")
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(9)
    acts = torch.randint(0, 3, (chunk, a.ld), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
    long_acts = acts.repeat(T // chunk, 1)               # the fused launch reads [T][ld]; the graph replays its 100 rows
    graph = b.capture_rollout(chunk, actions=acts, keep_all=False)
    episodes = 0
    for k in range(launches):
        t0 = time.time()
        ra, ca = a.rollout(T, actions=long_acts, fused=True, keep_all=False)
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(T // chunk):
            rb, cb = graph.launch()
        torch.cuda.synchronize()
        t2 = time.time()
        same = (torch.equal(a.state, b.state) and torch.equal(a.time, b.time) and torch.equal(ra[:n], rb[:n]) and torch.equal(ca[:n], cb[:n]))
        episodes += int((cb[:n] != 0).sum())
        print("launch %d: %d steps fused %.3f s (%.2f us/step), per-step graph %.3f s (%.2f us/step): %s" % (
            k, T, t1 - t0, (t1 - t0) * 1e6 / T, t2 - t1, (t2 - t1) * 1e6 / T, "EQUAL" if same else "DIFFERENT"), flush=True)
        if not same:
            sys.exit(1)
    st = a.state[:, :n]
    ok = bool(torch.isfinite(st).all()) and bool((st[5:7].abs() <= 0.05).all()) and bool((a.time[:n] >= -4).all()) and bool((a.time[:n] <= 1001).all())
    print("soak: %d x %d steps x %d worlds = %.2e world-steps, %s; invariants %s" % (launches, T, n, launches * T * n, mode, "hold" if ok else "BROKEN"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
