#!/usr/bin/env python3
"""wall-clock (s_memrealtime, 10 ns) stamps of the ring kernel (auto_reset 3): refill vs stepping blocks (diagnostic build:
AQUA_HIP_LIB=aquaticgymenv_amd/lib/variants/libaqua_hip_stamps.so).  MODE=2 stamps the scanning kernel instead."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from aquaticgymenv_amd import _capi, presets
from aquaticgymenv_amd.batched import BatchedAqua
n = int(os.environ.get("N", 262144))
mode = int(os.environ.get("MODE", 3))
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=mode, device="cuda:0")
env.reset()
MW, ROWS = 4, 4
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
for rep in range(3):
    stamps.zero_()
    env.rollout(1, actions=acts, keep_all=False)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64).reshape(R + M, MW, 8)
    rs, mn = s[:R], s[R:]
    t3 = s[:, :, 3].min()
    act = rs[:, :, 2] > 0
    print("mode %d launch %d: span %.2f us" % (mode, rep, us(max(mn[:, :, 2].max(), rs[:, :, 2][act].max()) - t3)))
    print("  starts after the first: refill/reseed %s | step %s" % (q(rs[:, :, 3] - t3), q(mn[:, :, 3] - t3)))
    print("  step  : consts %s" % q(mn[:, :, 0] - mn[:, :, 3]))
    print("          loads back+draws %s" % q(mn[:, :, 4] - mn[:, :, 3]))
    print("          arithmetic done %s" % q(mn[:, :, 1] - mn[:, :, 3]))
    print("          end %s   | end after first start %s" % (q(mn[:, :, 2] - mn[:, :, 3]), q(mn[:, :, 2] - t3)))
    print("  refill: consts %s" % q(rs[:, :, 0] - rs[:, :, 3]))
    print("          behind barrier %s" % q(rs[:, :, 1] - rs[:, :, 3]))
    print("          end %s   | end after first start %s" % (q((rs[:, :, 2] - rs[:, :, 3])[act]), q((rs[:, :, 2] - t3)[act])))
