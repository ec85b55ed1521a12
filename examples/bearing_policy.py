#!/usr/bin/env python3
"""The reference's hand-coded policy (main/testing/test_optimal.py: turn towards the goal, then full throttle) on a
batch of worlds, evaluated inside the step kernel: success rate over one episode per world.

    python examples/bearing_policy.py [--envs 100000] [--obstacles]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aquaticgymenv_amd.batched import BatchedAqua

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=100000)
ap.add_argument("--obstacles", action="store_true", help="the reference's default five obstacles")
args = ap.parse_args()

env = BatchedAqua(args.envs, obstacles=args.obstacles, seed=1, auto_reset=False)
env.reset()
first = torch.zeros(args.envs, dtype=torch.uint8, device=env.device)        # first termination code of every world
for step in range(1001):
    obs, reward, term = env.step(policy="bearing")                          # the action is computed on the device
    first = torch.where(first == 0, term, first)
    if step % 50 == 49 and int((first == 0).sum()) == 0:
        break
names = {1: "collided", 2: "time limit", 3: "reached the goal"}
for code, name in names.items():
    print("%-18s %6.2f %%" % (name, 100.0 * float((first == code).float().mean())))
