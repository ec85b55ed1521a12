"""Long replay of a captured 100-step rollout at 262 144 worlds, both action spaces; markers and state invariants at the end.
usage: python tools/soak.py [replays=1000]   (1 000 replays = 100 000 steps = 2.6e10 world-steps per action space)"""
import os, sys, time
REPLAYS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aquaticgymenv_amd import presets
from aquaticgymenv_amd.batched import BatchedAqua
for cont in (False, True):
    env = BatchedAqua(262144, obstacles=presets.BENCH8, seed=5, auto_reset="next_step", continuous=cont, device="cuda:0")
    env.reset()
    g = env.capture_rollout(100, actions="random", keep_all=False)
    t0 = time.time()
    for rep in range(REPLAYS):
        g.launch()
    torch.cuda.synchronize()
    n = env.num_envs
    st, tm = env.state[:, :n], env.time[:n]
    last = env._tick - 1
    neg = tm[tm < 0]
    ok = bool(((neg == -1 - (last & 1)) | (neg == -3 - (last & 1))).all()) and bool(torch.isfinite(st).all()) \
        and float(st[0:2].min()) >= 1.5 and float(st[0:2].max()) <= 98.5 and int(tm.max()) <= 1000
    print("continuous" if cont else "discrete", "%d steps in %.2f s, invariants %s, pending now %d" % (100 * REPLAYS, time.time() - t0, ok, int((tm < 0).sum())))
    assert ok
