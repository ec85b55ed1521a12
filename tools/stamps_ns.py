#!/usr/bin/env python3
"""wall-clock (s_memrealtime) stamps of the next-step kernel: main vs worker wavefronts (diagnostic build)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from aquaticgymenv_amd import _capi, presets
from aquaticgymenv_amd.batched import BatchedAqua
n = 262144
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
MW = int(os.environ.get('NS_MAIN', '8'))
blocks = (n + 64 * MW - 1) // (64 * MW)
stamps = torch.zeros((blocks * (MW + 2), 8), dtype=torch.int64, device="cuda")
_capi.lib.aqua_debug_set_stamps.argtypes = [ctypes.c_void_p]
_capi.check(_capi.lib.aqua_debug_set_stamps(stamps.data_ptr()), "set stamps")
acts = torch.randint(0, 3, (64, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
env.rollout(60, actions=acts, keep_all=False)
for rep in range(5):
    stamps.zero_()
    env.rollout(1, actions=acts, keep_all=False)
    torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64).reshape(blocks, MW + 2, 8)
t0 = s[:, :, 0][s[:, :, 0] > 0].min()
main, work = s[:, :MW, :], s[:, MW:, :]
def us(x): return x * 1e-2
print("kernel span (first start -> last end): %.2f us" % us(max(main[:, :, 2].max(), work[:, :, 2].max()) - t0))
print("main  : start %5.2f..%5.2f  loads+philox done median +%.2f  end median +%.2f  p99 +%.2f  max +%.2f  (latest end at %.2f)" % (
    us(main[:, :, 0].min() - t0), us(main[:, :, 0].max() - t0), us(np.median(main[:, :, 1] - main[:, :, 0])),
    us(np.median(main[:, :, 2] - main[:, :, 0])), us(np.percentile(main[:, :, 2] - main[:, :, 0], 99)), us((main[:, :, 2] - main[:, :, 0]).max()),
    us(main[:, :, 2].max() - t0)))
print("worker: start %5.2f..%5.2f  list known median +%.2f  end median +%.2f  p99 +%.2f  max +%.2f  (latest end at %.2f)" % (
    us(work[:, :, 0].min() - t0), us(work[:, :, 0].max() - t0), us(np.median(work[:, :, 1] - work[:, :, 0])),
    us(np.median(work[:, :, 2] - work[:, :, 0])), us(np.percentile(work[:, :, 2] - work[:, :, 0], 99)), us((work[:, :, 2] - work[:, :, 0]).max()),
    us(work[:, :, 2].max() - t0)))
w0 = work[:, 0, :]
print("worker 0 reseed (list known -> end): median %.2f us  p90 %.2f  max %.2f" % (us(np.median(w0[:, 2] - w0[:, 1])), us(np.percentile(w0[:, 2] - w0[:, 1], 90)), us((w0[:, 2] - w0[:, 1]).max())))
