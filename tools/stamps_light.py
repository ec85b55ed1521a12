#!/usr/bin/env python3
"""start/end wall-clock stamps (10 ns) of every wavefront of the next-step kernel (build with -DAQUA_STAMPS=2).
NS_MAIN / NS_SCAN_ROWS must match the build."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aquaticgymenv_amd import _capi, presets
from aquaticgymenv_amd.batched import BatchedAqua
n = int(os.environ.get("N", 262144))
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
MW = int(os.environ.get('NS_MAIN', '4'))
ROWS = int(os.environ.get('NS_SCAN_ROWS', '4'))
tile = 64 * MW
R = (n + ROWS * tile - 1) // (ROWS * tile)
M = (n + tile - 1) // tile
stamps = torch.zeros(((R + M) * MW, 8), dtype=torch.int64, device="cuda")
_capi.lib.aqua_debug_set_stamps.argtypes = [ctypes.c_void_p]
_capi.check(_capi.lib.aqua_debug_set_stamps(stamps.data_ptr()), "set stamps")
acts = torch.randint(0, 3, (64, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
env.rollout(60, actions=acts, keep_all=False)
def us(x): return x * 1e-2
def q(x): return "median %.2f p90 %.2f p99 %.2f max %.2f" % (us(np.median(x)), us(np.percentile(x, 90)), us(np.percentile(x, 99)), us(x.max()))
for rep in range(4):
    stamps.zero_()
    env.rollout(1, actions=acts, keep_all=False)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64).reshape(R + M, MW, 8)
    rs, mn = s[:R], s[R:]
    t0 = s[:, :, 3].min()
    act = rs[:, :, 2] > 0
    print("launch %d: span %.2f us | starts: re-seed %s ; step %s" % (rep, us(max(mn[:, :, 2].max(), rs[:, :, 2].max()) - t0), q(rs[:, :, 3] - t0), q(mn[:, :, 3] - t0)))
    print("   life: step %s ; re-seed (working wavefronts, %d) %s" % (q(mn[:, :, 2] - mn[:, :, 3]), act.sum(), q((rs[:, :, 2] - rs[:, :, 3])[act])))
    print("   ends after first start: step %s ; re-seed %s" % (q(mn[:, :, 2] - t0), q((rs[:, :, 2] - t0)[act])))
    # step wavefronts by dispatch order: are the late finishers the late starters?
    order = np.argsort(mn[:, :, 3].reshape(-1))
    st, en = mn[:, :, 3].reshape(-1)[order] - t0, mn[:, :, 2].reshape(-1)[order] - t0
    k = len(order) // 8
    print("   step wavefronts in start order, octiles: start " + " ".join("%.2f" % us(st[i * k:(i + 1) * k].mean()) for i in range(8)) +
          " | life " + " ".join("%.2f" % us((en - st)[i * k:(i + 1) * k].mean()) for i in range(8)))
