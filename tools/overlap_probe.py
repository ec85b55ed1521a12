"""Do kernels of two HIP streams overlap on this part when one of them is the step launch?

A: a captured rollout of the step kernel (262 144 worlds).  B: a captured run of sparse masked resets (the re-seeding
routine of ~2 % of the worlds, the work a launch-decoupled refill of restart states would do).  Timed alone and
together (wall = first start .. last end).  No overlap: together = A + B; full overlap: max(A, B).
usage: python tools/overlap_probe.py [N] [steps per graph] [replays]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd import _capi                      # noqa: E402
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402
from aquaticgymenv_amd import presets                    # noqa: E402


def capture_resets(env, mask, count, stream):
    handle = ctypes.c_void_p()
    with torch.cuda.stream(stream):
        s = env._stream()
        _capi.check(_capi.lib.aqua_graph_begin(s), "begin")
        try:
            for _ in range(count):
                env.reset(mask)
        finally:
            _capi.check(_capi.lib.aqua_graph_end(s, ctypes.byref(handle)), "end")
    return handle


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    replays = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    obst = presets.BENCH8
    for mode in ("next_step", False):
        a = BatchedAqua(n, obstacles=obst, device=dev, seed=1, auto_reset=mode)
        b = BatchedAqua(n, obstacles=obst, device=dev, seed=2, auto_reset=False)
        a.reset()
        b.reset()
        acts = torch.randint(0, 9, (steps, n), dtype=torch.uint8, device=dev)
        ga = a.capture_rollout(steps, actions=acts)
        mask = (torch.rand(n, device=dev) < 0.02).to(torch.uint8)
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        gb = capture_resets(b, mask, steps, sb)
        torch.cuda.synchronize()

        def run(do_a, do_b):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            torch.cuda.synchronize()
            if do_a:
                with torch.cuda.stream(sa):
                    ev[0].record()
                    for _ in range(replays):
                        ga.launch()
                    ev[1].record()
            if do_b:
                with torch.cuda.stream(sb):
                    ev[2].record()
                    for _ in range(replays):
                        _capi.check(_capi.lib.aqua_graph_launch(gb, b._stream()), "launch")
                    ev[3].record()
            torch.cuda.synchronize()
            ta = ev[0].elapsed_time(ev[1]) if do_a else 0.0
            tb = ev[2].elapsed_time(ev[3]) if do_b else 0.0
            wall = max(ta, tb)
            if do_a and do_b:
                first = ev[0] if ev[0].elapsed_time(ev[2]) >= 0 else ev[2]
                last = ev[1] if ev[3].elapsed_time(ev[1]) >= 0 else ev[3]
                wall = first.elapsed_time(last)
            return ta, tb, wall

        for _ in range(2):
            run(True, True)
        k = steps * replays
        for name, da, db in (("A alone", True, False), ("B alone", False, True), ("together", True, True),
                             ("A alone", True, False), ("together", True, True)):
            ta, tb, wall = run(da, db)
            print("mode %-10s %-9s A %.3f us/step  B %.3f us/reset  wall %.3f us per pair" %
                  (mode, name, 1e3 * ta / k, 1e3 * tb / k, 1e3 * wall / k), flush=True)
        _capi.lib.aqua_graph_destroy(gb)
        ga.close()


if __name__ == "__main__":
    main()
