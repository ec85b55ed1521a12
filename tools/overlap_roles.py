"""The two roles of the next-step kernel as TWO kernels on two HIP streams: what would a launch-decoupled re-seeding
(restart states prepared ahead of the step that consumes them) be worth?

Needs the diagnostic builds `nw` (stepping blocks alone) and `nm` (re-seeding blocks alone, synthetic pending set) of
`python -m aquaticgymenv_amd.build --variants`.  S = a step launch of the nw build, R = one of the nm build.
Timed, per step: S alone, R alone, the shipped kernel (both roles in one launch), S and R as independent graphs on two
streams, and ONE graph with the edges a decoupled design needs (S_t -> R_t, R_t -> S_{t+lag}, S_t -> S_{t+1}).
usage: python tools/overlap_roles.py [N] [steps per graph] [replays]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd import _capi                      # noqa: E402
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402
from aquaticgymenv_amd import presets                    # noqa: E402

VARIANTS = os.path.join(os.path.dirname(os.path.abspath(_capi.__file__)), "lib", "variants")


def load(name):
    _capi.LIB_PATH = os.path.join(VARIANTS, "libaqua_hip_%s.so" % name)
    return _capi._load()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    replays = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    shipped = _capi.lib
    libs = {"both": shipped, "S": load("nw"), "R": load("nm")}
    envs = {}
    for i, k in enumerate(libs):
        _capi.lib = shipped
        envs[k] = BatchedAqua(n, obstacles=presets.BENCH8, device=dev, seed=i, auto_reset="next_step")
        envs[k].reset()
    act = torch.randint(0, 9, (n,), dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def step(k):
        _capi.lib = libs[k]
        envs[k].step(act)
        _capi.lib = shipped

    def capture(fn, stream):
        handle = ctypes.c_void_p()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            s = ctypes.c_void_p(stream.cuda_stream)
            _capi.check(shipped.aqua_graph_begin(s), "begin")
            try:
                fn()
            finally:
                _capi.check(shipped.aqua_graph_end(s, ctypes.byref(handle)), "end")
        return handle

    def chain(k):
        def fn():
            for _ in range(steps):
                step(k)
        return fn

    def coupled(lag):
        """S on s1, R on s2, one graph: R_t after S_t; S_t after S_{t-1} and R_{t-lag}"""
        def fn():
            done_r = []
            for t in range(steps):
                if t - lag >= 0:
                    s1.wait_event(done_r[t - lag])
                with torch.cuda.stream(s1):
                    step("S")
                    e = torch.cuda.Event()
                    e.record(s1)
                s2.wait_event(e)
                with torch.cuda.stream(s2):
                    step("R")
                    f = torch.cuda.Event()
                    f.record(s2)
                done_r.append(f)
            for f in done_r[-(lag + 1):]:
                s1.wait_event(f)
        return fn

    graphs = {k: capture(chain(k), s1 if k != "R" else s2) for k in libs}
    for lag in (1, 2, 4):
        try:
            graphs["coupled lag %d" % lag] = capture(coupled(lag), s1)
        except Exception as exc:                                    # noqa: BLE001
            print("coupled lag %d: capture failed: %s" % (lag, exc), flush=True)
    torch.cuda.synchronize()

    def launch(k, stream):
        _capi.check(shipped.aqua_graph_launch(graphs[k], ctypes.c_void_p(stream.cuda_stream)), "launch")

    def timed(jobs):
        ev = []
        torch.cuda.synchronize()
        for k, stream in jobs:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev.append((a, b))
        for (k, stream), (a, b) in zip(jobs, ev):
            a.record(stream)
        for _ in range(replays):
            for k, stream in jobs:
                launch(k, stream)
        for (k, stream), (a, b) in zip(jobs, ev):
            b.record(stream)
        torch.cuda.synchronize()
        per = [1e3 * a.elapsed_time(b) / (steps * replays) for a, b in ev]
        wall = max(1e3 * a.elapsed_time(b2) / (steps * replays) for a, _ in ev for _, b2 in ev)
        return per, wall

    plan = [[("both", s1)], [("S", s1)], [("R", s2)], [("S", s1), ("R", s2)]]
    plan += [[(k, s1)] for k in graphs if k.startswith("coupled")]
    for rep in range(3):
        for jobs in plan:
            per, wall = timed(jobs)
            print("%-28s per step %s   wall %.3f us" % (" + ".join(k for k, _ in jobs),
                                                        " ".join("%.3f" % p for p in per), wall), flush=True)
        print(flush=True)


if __name__ == "__main__":
    main()
