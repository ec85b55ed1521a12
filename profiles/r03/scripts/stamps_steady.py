#!/usr/bin/env python3
"""start/end wall-clock stamps (s_memrealtime, 10 ns) of every wavefront of the LAST launch of a replayed 20-step graph --
the steady-state launch, not a cold eager one (build with -DAQUA_STAMPS=2: libaqua_hip_stamps2.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aquaticgymenv_amd import _capi, presets
from aquaticgymenv_amd.batched import BatchedAqua
n = 262144
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
MW, ROWS = 4, 4
tile = 64 * MW
R, M = n // (ROWS * tile), n // tile
stamps = torch.zeros(((R + M) * MW, 8), dtype=torch.int64, device="cuda")
_capi.lib.aqua_debug_set_stamps.argtypes = [ctypes.c_void_p]
_capi.check(_capi.lib.aqua_debug_set_stamps(stamps.data_ptr()), "set stamps")
acts = torch.randint(0, 3, (100, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
g = env.capture_rollout(20, actions=acts, keep_all=False)
for _ in range(5):
    g.launch()
def us(x): return x * 1e-2
def q(x): return "p1 %.2f p10 %.2f median %.2f p90 %.2f p99 %.2f max %.2f" % tuple(us(np.percentile(x, p)) for p in (1, 10, 50, 90, 99, 100))
for rep in range(4):
    for _ in range(3):
        g.launch()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64).reshape(R + M, MW, 8)
    rs, mn = s[:R], s[R:]
    t0 = s[:, :, 3].min()
    act = rs[:, :, 2] > 0
    end = max(mn[:, :, 2].max(), rs[:, :, 2][act].max())
    print("launch span (first wavefront start -> last wavefront end): %.2f us" % us(end - t0))
    print("  starts : re-seed %s | step %s" % (q(rs[:, :, 3] - t0), q(mn[:, :, 3] - t0)))
    print("  life   : step %s | re-seed (working, %d) %s" % (q(mn[:, :, 2] - mn[:, :, 3]), act.sum(), q((rs[:, :, 2] - rs[:, :, 3])[act])))
    print("  ends   : step %s | re-seed %s" % (q(mn[:, :, 2] - t0), q((rs[:, :, 2] - t0)[act])))
    # by block index (dispatch order): mean end of the step blocks in 8 consecutive groups
    be = (mn[:, :, 2].max(axis=1) - t0)
    k = M // 8
    print("  step block ends by block index, octiles: " + " ".join("%.2f" % us(be[i * k:(i + 1) * k].mean()) for i in range(8)) +
          " | worst blocks: " + " ".join("%d" % i for i in np.argsort(be)[-6:]))
