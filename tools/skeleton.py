#!/usr/bin/env python3
"""What a launch of the step's shape costs without the step: empty body, memory traffic only, and memory traffic
plus a dependent-FMA chain (diagnostic build).  100 launches per graph, like bench.py.
AQUA_HIP_LIB=aquaticgymenv_amd/lib/variants/libaqua_hip_stamps.so python tools/skeleton.py [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd import _capi
lib = _capi.lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
f = lib.aqua_debug_skeleton
f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
lib.aqua_graph_begin.argtypes = [ctypes.c_void_p]
lib.aqua_graph_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
lib.aqua_graph_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros(11 * n, dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()
# kinds 3-7 (round 3): the state as 32-byte records (3, 5: a record per lane; 6, 7: consecutive 16-byte pieces per lane and an
# LDS transposition), and the layouts with ~2 % of the worlds restarting (4, 5, 7)
for kind, block, spin in ((0, 512, 0), (0, 256, 0), (1, 512, 0), (1, 256, 0), (2, 512, 1), (3, 256, 0), (4, 256, 0), (5, 256, 0),
                          (6, 256, 0), (7, 256, 0), (1, 256, 0), (4, 256, 0), (6, 256, 0), (7, 256, 0)):
    with torch.cuda.stream(stream):
        s = stream.cuda_stream
        _capi.check(lib.aqua_graph_begin(s), "begin")
        for _ in range(100):
            _capi.check(f(kind, block, buf.data_ptr(), n, n, spin, s), "launch")
        g = ctypes.c_void_p()
        _capi.check(lib.aqua_graph_end(s, ctypes.byref(g)), "end")
        for _ in range(3):
            lib.aqua_graph_launch(g, s)
        stream.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                lib.aqua_graph_launch(g, s)
            e1.record(stream)
            stream.synchronize()
            best = min(best, e0.elapsed_time(e1))
    print("kind %d block %4d spin %4d: %.2f us per launch" % (kind, block, spin, best))
