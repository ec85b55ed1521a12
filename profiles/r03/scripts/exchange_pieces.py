#!/usr/bin/env python3
"""Which piece of the done-mask exchange costs the step stream its microsecond?  2 000 steps (20 graphs of 100) of 262 144
worlds on the launch stream per measurement; every 500 steps one 16 MB done-mask block; the pieces of
DoneMaskExchange.gather_async / wait_source added one at a time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aquaticgymenv_amd import presets, _capi
from aquaticgymenv_amd.batched import BatchedAqua

dev = torch.device("cuda:0")
launch = torch.cuda.Stream(device=dev)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(launch)
env = BatchedAqua(262144, obstacles=presets.BENCH8, seed=0, auto_reset="next_step", device=dev)
env.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = torch.randint(0, 3, (100, env.ld), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
words = env.ld // 64
hist = [torch.zeros((500, words), dtype=torch.int64, device=dev) for _ in range(2)]
recv = torch.zeros((4, 500, words), dtype=torch.int64, device=dev)
graphs = {(b, r): env.capture_rollout(100, actions=acts, keep_all=False, done_history=hist[b][r:r + 100]) for b in range(2) for r in range(0, 500, 100)}
lib = _capi.lib


def region(piece):
    busy = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(launch)
    for blk in range(4):
        b = blk & 1
        if piece >= 5 and b in busy:
            launch.wait_event(busy.pop(b))                      # wait_source
        for r in range(0, 500, 100):
            graphs[(b, r)].launch()
        if piece >= 1:
            ev = launch.record_event()
        if piece >= 2:
            side.wait_event(ev)
        if piece == 3:
            with torch.cuda.stream(side):
                recv[blk].copy_(hist[b], non_blocking=True)     # torch's copy kernel
        if piece >= 4:
            _capi.check(lib.aqua_copy_async(recv[blk].data_ptr(), hist[b].data_ptr(), 500 * words * 8, 0,
                                            ctypes.c_void_p(side.cuda_stream)), "copy")
        if piece >= 5:
            busy[b] = side.record_event()
        if piece >= 6:
            hist[b].record_stream(side)
    e1.record(launch)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 2000


names = ["no exchange", "+ event recorded on the launch stream per block", "+ side stream waits for it", "+ torch copy_ of the block on the side stream",
         "+ aqua_copy_async (wavefronts) instead", "+ side event, launch stream waits for it before reusing the buffer", "+ record_stream"]
for rep in range(2):
    for piece, name in enumerate(names):
        region(piece)
        v = sorted(region(piece) for _ in range(3))
        print("%-72s %.3f us per step" % (name, v[1]), flush=True)
