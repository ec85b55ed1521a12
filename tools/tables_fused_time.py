"""per-world obstacle tables (8 rows): one launch per step (captured graph) against the fused rollout, us per step.
usage: python tools/tables_fused_time.py [N] [steps] [rows per world]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402
from aquaticgymenv_amd import presets                    # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.RandomState(7)
    if K == 8:
        tables = np.repeat(presets.BENCH8[None], n, axis=0).astype(np.float64)
        tables[:, :, 0:2] += rng.uniform(-3, 3, (n, 8, 2))
    else:                                                   # K rows per world, sized so that ~a quarter of a world is blocked
        tables = np.zeros((n, K, 5))                        # (boat radius included): mean radius sqrt(0.28 * 100^2 / (pi K)) - 2.5
        tables[:, :, 0:2] = rng.uniform(10, 90, (n, K, 2))
        kind = rng.randint(0, 2, (n, K)).astype(np.float64)
        scale = max(0.05, ((0.28 * 1.0e4 / (np.pi * K)) ** 0.5 - 2.5) / 6.0)
        tables[:, :, 2] = kind
        tables[:, :, 3] = np.where(kind == 0, rng.uniform(2, 10, (n, K)), rng.uniform(5, 15, (n, K))) * scale
        tables[:, :, 4] = np.where(kind == 0, 0.0, rng.uniform(5, 15, (n, K)) * scale)
    print("%d worlds, %d rows per world" % (n, K), flush=True)
    acts = torch.randint(0, 3, (steps, n), dtype=torch.uint8, device="cuda:0")
    for mode in ("next_step", "same_step", False):
        for fused in (False, True):
            env = BatchedAqua(n, obstacles=tables, device="cuda:0", seed=3, auto_reset=mode)
            env.reset()
            g = env.capture_rollout(steps, actions=acts, fused=fused)
            for _ in range(3):
                g.launch()
            torch.cuda.synchronize()
            best = []
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    g.launch()
                b.record()
                torch.cuda.synchronize()
                best.append(1e3 * a.elapsed_time(b) / (5 * steps))
            print("%-10s %-22s %s us/step" % (mode, "fused (one launch)" if fused else "one launch per step", " ".join("%.3f" % v for v in best)), flush=True)


if __name__ == "__main__":
    main()
