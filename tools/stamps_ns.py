#!/usr/bin/env python3
"""wall-clock (s_memrealtime, 10 ns) stamps of the next-step kernel: re-seeding vs stepping blocks (diagnostic build).
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
MW = int(os.environ.get('NS_MAIN', '6'))
ROWS = int(os.environ.get('NS_SCAN_ROWS', '4'))
tile = 64 * MW
R = (n + ROWS * tile - 1) // (ROWS * tile)
M = (n + tile - 1) // tile
stamps = torch.zeros(((R + M) * MW, 8), dtype=torch.int64, device="cuda")
_capi.lib.aqua_debug_set_stamps.argtypes = [ctypes.c_void_p]
_capi.check(_capi.lib.aqua_debug_set_stamps(stamps.data_ptr()), "set stamps")
acts = torch.randint(0, 3, (64, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
env.rollout(60, actions=acts, keep_all=False)
for rep in range(5):
    stamps.zero_()
    env.rollout(1, actions=acts, keep_all=False)
    torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64).reshape(R + M, MW, 8)
rs, mn = s[:R], s[R:]
t3 = s[:, :, 3].min()
def us(x): return x * 1e-2
def q(x): return "median %.2f p90 %.2f p99 %.2f max %.2f" % (us(np.median(x)), us(np.percentile(x, 90)), us(np.percentile(x, 99)), us(x.max()))
print("R=%d re-seeding blocks, M=%d stepping blocks, %d wavefronts each" % (R, M, MW))
print("wavefront starts after the first one: re-seed %s | step %s" % (q(rs[:, :, 3] - t3), q(mn[:, :, 3] - t3)))
print("step   : from own start: consts %s" % q(mn[:, :, 0] - mn[:, :, 3]))
print("         loads back+draws %s" % q(mn[:, :, 4] - mn[:, :, 3]))
print("         end %s" % q(mn[:, :, 2] - mn[:, :, 3]))
print("         end after first start: %s" % q(mn[:, :, 2] - t3))
act = rs[:, :, 2] > 0
print("re-seed: from own start: consts %s" % q(rs[:, :, 0] - rs[:, :, 3]))
print("         list known %s" % q(rs[:, :, 1] - rs[:, :, 3]))
print("         end %s" % q((rs[:, :, 2] - rs[:, :, 3])[act]))
print("         end after first start: %s" % q((rs[:, :, 2] - t3)[act]))
print("         re-seed loop (list known -> end): %s" % q((rs[:, :, 2] - rs[:, :, 1])[act]))
