#!/usr/bin/env python3
"""What shader clock do the step kernels run at?  A probe kernel (tools/micro/clock_probe.hip: delta s_memtime / delta
s_memrealtime x 100 MHz around a dependent fma chain) launched (a) on an idle GPU, (b) directly behind 10 replays of the
100-step graph of 262 144 worlds, (c) behind ONE 20-step graph that started on an idle GPU (the driver's shape).
build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/libclock_probe.so tools/micro/clock_probe.hip"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else "/tmp/libclock_probe.so")
lib.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
env = BatchedAqua(262144, obstacles=presets.BENCH8, seed=0, auto_reset="next_step", device=dev)
env.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = torch.randint(0, 3, (100, env.ld), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
g100 = env.capture_rollout(100, actions=acts, keep_all=False)
g20 = env.capture_rollout(20, actions=acts, keep_all=False)
out = torch.zeros(256 * 4, dtype=torch.int64, device=dev)


def probe(label, before, iters=2000):
    vals = []
    for _ in range(5):
        torch.cuda.synchronize()
        time.sleep(0.002)
        before()
        lib.clock_probe(out.data_ptr(), 256, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(256, 4).astype(np.float64)
        ghz = o[:, 0] / o[:, 1] * 0.1
        vals.append(np.median(ghz))
        span = (o[:, 3].max() - o[:, 2].min()) * 1e-2
    print("%-52s clock GHz: %s   (chain of %d fma: %.0f cycles = %.2f cyc/fma; probe span %.1f us)" %
          (label, " ".join("%.2f" % v for v in vals), iters, np.median(o[:, 0]), np.median(o[:, 0]) / iters, span), flush=True)


probe("idle GPU", lambda: None)
probe("behind 10 x 100-step graph replays", lambda: [g100.launch() for _ in range(10)])
probe("behind 100 x 100-step graph replays (50 ms)", lambda: [g100.launch() for _ in range(100)])
probe("behind ONE 20-step graph from idle", lambda: g20.launch())
probe("behind 5-step + 20-step graphs from idle", lambda: (g20.launch(), g20.launch()))
probe("idle GPU again, long chain", lambda: None, iters=20000)
