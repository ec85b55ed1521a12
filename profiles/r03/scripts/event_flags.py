#!/usr/bin/env python3
"""Does the kind of HIP event change what a 20-step region reads?  The same graph between two events created with
hipEventDefault, hipEventDisableSystemFence, hipEventReleaseToDevice and hipEventReleaseToSystem (hipEventCreateWithFlags
through ctypes on the runtime torch has loaded), next to torch.cuda.Event."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua

hip = ctypes.CDLL("libamdhip64.so")
vp = ctypes.c_void_p
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
hip.hipEventRecord.argtypes = [vp, vp]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), vp, vp]
hip.hipEventSynchronize.argtypes = [vp]

n, K = 262144, 20
env = BatchedAqua(n, obstacles=presets.BENCH8, seed=0, auto_reset=2, device="cuda:0")
env.reset()
acts = torch.randint(0, 3, (100, env.ld), device="cuda", dtype=torch.int64).to(torch.uint8)
g = env.capture_rollout(K, actions=acts, keep_all=False)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    g.launch()
torch.cuda.synchronize()


def make(flags):
    e = vp()
    rc = hip.hipEventCreateWithFlags(ctypes.byref(e), flags)
    if rc:
        raise RuntimeError("hipEventCreateWithFlags(0x%x) -> %d" % (flags, rc))
    return e


KINDS = (("torch.cuda.Event", None), ("hipEventDefault", 0x0), ("hipEventDisableSystemFence", 0x20000000),
         ("hipEventReleaseToDevice", 0x40000000), ("hipEventReleaseToSystem", 0x80000000))
for rep in range(2):
    for name, flags in KINDS:
        vals = []
        for _ in range(15):
            torch.cuda.synchronize()
            if flags is None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.launch(); e1.record()
                torch.cuda.synchronize()
                vals.append(e0.elapsed_time(e1) * 1e3 / K)
            else:
                e0, e1 = make(flags), make(flags)
                hip.hipEventRecord(e0, stream); g.launch(); hip.hipEventRecord(e1, stream)
                torch.cuda.synchronize()
                ms = ctypes.c_float()
                rc = hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
                vals.append(ms.value * 1e3 / K if rc == 0 else float("nan"))
        vals.sort()
        print("%-28s 20-step region, us per step: median %.3f min %.3f max %.3f" % (name, vals[7], vals[0], vals[-1]), flush=True)
