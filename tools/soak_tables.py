"""long bit-for-bit comparison: shared-table kernels == per-world kernels (every world holding the same list, rows handed
over through LDS) == fused per-world rollout, device-sampled actions.  usage: python tools/soak_tables.py [N] [steps] [rows]
rows > 8: BENCH8 + (rows - 8) small seeded obstacles -- the paths of tables too long for registers (round 5: rows left in LDS
by the lane that streams them, the world-major copy, the split re-seeding pass) against the shared-table kernels' row loops"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402
from aquaticgymenv_amd import presets                    # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000 + 37
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    base = np.asarray(presets.BENCH8, dtype=np.float64)
    if rows > 8:
        rng = np.random.RandomState(rows)
        extra = np.zeros((rows - 8, 5))
        extra[:, 0:2] = rng.uniform(8, 92, (rows - 8, 2))
        extra[:, 2] = rng.randint(0, 2, rows - 8)
        extra[:, 3] = rng.uniform(0.5, 2.5, rows - 8)
        extra[:, 4] = np.where(extra[:, 2] == 1, rng.uniform(0.5, 2.5, rows - 8), 0.0)
        base = np.concatenate([base, extra])
    tables = np.repeat(base[None], n, axis=0).astype(np.float64)
    for mode in ("next_step", "same_step"):
        envs = {"shared": BatchedAqua(n, obstacles=base, device="cuda:0", seed=2024, auto_reset=mode),
                "per-world": BatchedAqua(n, obstacles=tables, device="cuda:0", seed=2024, auto_reset=mode),
                "per-world fused": BatchedAqua(n, obstacles=tables, device="cuda:0", seed=2024, auto_reset=mode)}
        for e in envs.values():
            e.reset()
        finished = 0
        for t0 in range(0, steps, 100):
            out = {k: e.rollout(100, fused=(k == "per-world fused"), keep_all=True) for k, e in envs.items()}
            ref = envs["shared"]
            for k, e in envs.items():
                assert torch.equal(e.state[:, :n], ref.state[:, :n]) and torch.equal(e.time[:n], ref.time[:n]), (mode, k, t0)
                assert torch.equal(out[k][0][:, :n], out["shared"][0][:, :n]) and torch.equal(out[k][1][:, :n], out["shared"][1][:, :n]), (mode, k, t0)
            finished += int((out["shared"][1][:, :n] != 0).sum().item())
        print("%s, %d rows: %d steps x %d worlds, %d episodes finished: shared == per-world == fused per-world, bit for bit" % (mode, rows, steps, n, finished), flush=True)


if __name__ == "__main__":
    main()
