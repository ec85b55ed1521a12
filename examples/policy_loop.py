#!/usr/bin/env python3
"""The reference's evaluation loop (main/testing/__init__.py:17-36: reset, then get_action -> step until done) for a batch of
worlds with the policy in PyTorch and the step as ONE captured HIP graph: the policy writes its actions into a fixed device
buffer, the graph re-reads the buffer at every replay -- no per-step marshalling on the host.

    python examples/policy_loop.py [worlds=65536] [steps=400]

The policy here is the reference's hand-coded bearing rule (main/testing/test_optimal.py:8-28) written with torch ops on
the observation view (the kernels also have it built in: actions="bearing"); replace `policy()` by a network."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aquaticgymenv_amd import presets                    # noqa: E402
from aquaticgymenv_amd.batched import BatchedAqua       # noqa: E402


THRESHOLD = 8.0 / 180.0 * math.pi                        # OptimalAquaPolicy.ANGLE_THRESHOLD


def policy(obs):
    """obs [N][5] = x, y, theta, goal_x, goal_y (a view of the state rows) -> action index [N]: the rule of
    main/testing/test_optimal.py:16-28 -- both angles folded into [0, 2 pi), their plain difference (not wrapped, as
    there), turn left (0) / right (1) outside the threshold, straight (2) inside"""
    x, y, th, gx, gy = obs.unbind(1)
    two_pi = 2.0 * math.pi
    boat = torch.remainder(th + math.pi / 2 + two_pi, two_pi)
    goal = torch.remainder(torch.atan2(gy - y, gx - x) + two_pi, two_pi)
    diff = goal - boat
    return torch.where(diff.abs() > THRESHOLD, torch.where(diff > 0, 0, 1), 2).to(torch.int64)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    env = BatchedAqua(n, obstacles=presets.DEFAULT5, seed=0, auto_reset="next_step", device="cuda:0")
    obs = env.reset()
    action = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    step = env.capture_step(action)                      # aqua_step_f32 + the tick advance, captured once
    ended = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    for _ in range(steps):
        action.copy_(policy(obs))                        # the graph reads the buffer as it is at replay time
        reward, term = step.launch()
        ended += torch.bincount(term[:n].to(torch.int64), minlength=4)
    torch.cuda.synchronize()
    none, collided, timed_out, success = (int(v) for v in ended)
    episodes = collided + timed_out + success
    print("%d worlds x %d steps: %d episodes ended -- %.1f %% at the goal, %.1f %% collided, %.1f %% out of time"
          % (n, steps, episodes, 100.0 * success / max(episodes, 1), 100.0 * collided / max(episodes, 1), 100.0 * timed_out / max(episodes, 1)))


if __name__ == "__main__":
    main()
