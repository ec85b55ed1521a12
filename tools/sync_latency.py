"""wall clock of one 20-step graph replay on an idle GPU, by the way the host waits for it (us, median of 200):
torch.cuda.synchronize / stream.synchronize / event.synchronize / polling event.query()"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402
from aquaticgymenv_amd import presets                    # noqa: E402

n, T = 262144, int(sys.argv[1]) if len(sys.argv) > 1 else 20
env = BatchedAqua(n, obstacles=presets.BENCH8, device="cuda:0", seed=0, auto_reset="next_step")
env.reset()
acts = torch.randint(0, 3, (T, n), dtype=torch.uint8, device="cuda:0")
g = env.capture_rollout(T, actions=acts, timing=True)
stream = torch.cuda.current_stream()
for _ in range(10):
    g.launch()
torch.cuda.synchronize()


def wait_device():
    torch.cuda.synchronize()


def wait_stream():
    stream.synchronize()


def wait_event():
    e = torch.cuda.Event()
    e.record()
    e.synchronize()


def wait_poll():
    e = torch.cuda.Event()
    e.record()
    while not e.query():
        pass


for name, wait in (("torch.cuda.synchronize", wait_device), ("stream.synchronize", wait_stream),
                   ("event.synchronize", wait_event), ("event.query polling", wait_poll)) * 2:
    walls, gpu = [], []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.launch()
        wait()
        walls.append(time.perf_counter() - t0)
        gpu.append(g.elapsed_ms())
    walls.sort(); gpu.sort()
    print("%-24s wall median %.1f us  p10 %.1f  | graph nodes %.1f us" % (name, 1e6 * walls[100], 1e6 * walls[20], 1e3 * gpu[100]), flush=True)
