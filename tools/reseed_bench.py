#!/usr/bin/env python3
"""cycles of one warm re-seeding call / one Philox draw (diagnostic build).
AQUA_HIP_LIB=aquaticgymenv_amd/lib/variants/libaqua_hip_stamps.so python tools/reseed_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd import _capi, presets
for name, rows in (("bench8", presets.BENCH8), ("none", presets.NONE)):
    blob = _capi.pack_obstacles(rows)
    dev = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda() if blob else None
    out = torch.zeros(8, dtype=torch.int64, device="cuda")
    p = _capi.AquaParams(waves=1, random_boat=1, random_goal=1, time_limit=1000)
    f = _capi.lib.aqua_debug_reseed_bench
    f.argtypes = [ctypes.POINTER(_capi.AquaParams), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    _capi.check(f(ctypes.byref(p), dev.data_ptr() if dev is not None else None, rows.shape[0], out.data_ptr(), None), "bench")
    torch.cuda.synchronize()
    print(name, "group re-seeding: %d cycles/call   philox draw: %d cycles   pair draw: %d cycles   exact_step 1st/2nd/3rd call: %d / %d / %d cycles" %
          (int(out[0]), int(out[1]), int(out[6]), int(out[3]), int(out[4]), int(out[5])))
