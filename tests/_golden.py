"""helpers to read tests/golden/*.npz (data generated from the reference by make_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class StepGolden(object):
    def __init__(self):
        z = np.load(os.path.join(GOLDEN, "step_golden.npz"))
        self.z = z
        self.n_cfg = int(z["n_cfg"])
        self.n_hand = int(z["n_hand"])
        self.n = z["cfg"].shape[0]

    def cfg(self, ci):
        z = self.z
        return dict(obstacles=z["cfg%d_obstacles" % ci].reshape(-1, 5), continuous=bool(z["cfg%d_continuous" % ci]),
                    waves=int(z["cfg%d_waves" % ci]), name=str(z["cfg%d_name" % ci]))

    def rows(self, ci):
        """all rows of one configuration as a dict of arrays (SoA-ready)."""
        z = self.z
        idx = np.nonzero(z["cfg"] == ci)[0]
        out = {k: z[k][idx] for k in ("state_in", "time_in", "action_i", "action_c", "noise_u", "pose", "reward",
                                     "term", "wave_out", "thrust_total", "m_border", "m_obst", "m_goal", "time_out",
                                     "reward_is_int")}
        out["index"] = idx
        return out


def load_traj():
    return np.load(os.path.join(GOLDEN, "traj_golden.npz"))


def load_episodes():
    return np.load(os.path.join(GOLDEN, "episodes_golden.npz"))


def load_reset():
    return np.load(os.path.join(GOLDEN, "reset_golden.npz"))


def angle_diff(a, b):
    """|a - b| modulo 2 pi."""
    d = (np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64) + np.pi) % (2 * np.pi) - np.pi
    return np.abs(d)
