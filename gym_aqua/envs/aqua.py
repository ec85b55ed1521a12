"""gym_aqua.envs.AquaEnv / AquaContinuousEnv -- the reference's Gym-shaped API over the batched HIP
path (drop-in for gym_aqua/envs/aqua.py:9-213, 458-459 of ilVecc/AquaticGymEnv).

Constructor keywords of the reference (aqua.py:13) are kept: obstacles, waves, random_boat,
random_goal.  Three more select the batch: num_envs (default 1), device, seed.

num_envs == 1  -> reference-shaped returns: reset() -> float64[5]; step(a) -> (float64[5], reward,
                  bool, {'Termination.collided', 'Termination.time', 'Termination.success'}), with the
                  reference's types (terminal rewards are the Python ints +-10, aqua.py:86-88); no
                  auto-reset, no freeze after done (the caller resets, as main/impl/dqn.py:150 does).
num_envs  > 1  -> device tensors: obs [N,5] float32 (a strided VIEW of the state rows), reward [N]
                  float32, done [N] bool, info = the same three keys holding bool tensors [N]
                  (plus 'term': uint8 codes).  Finished worlds restart by themselves; `autoreset` picks the
                  convention: "next_step" (default, Gymnasium >= 1.0: the world restarts during the NEXT
                  step, which returns the fresh observation with reward 0 / done False; fastest) or
                  "same_step" (classic gym VectorEnv: the returned observation is already the new episode's).
All arithmetic happens in libaqua_hip.so; there is no CPU implementation behind this class.
"""
import math

import numpy as np

from aquaticgymenv_amd import presets, spaces
from aquaticgymenv_amd.batched import BatchedAqua, TIME_LIMIT

try:                                   # classic gym is optional (absent from the build image); gymnasium's Env base is
    import gym as _gym                 # not used: these classes speak the reference's gym 0.17 protocol
    _EnvBase = _gym.Env
except Exception:                      # pragma: no cover - depends on the environment
    _gym = None
    _EnvBase = object

INFO_KEYS = ("Termination.collided", "Termination.time", "Termination.success")


def space_types(gym_module=None):
    """(Box, Discrete): classic gym's own classes when gym is importable -- so that `isinstance(env.action_space,
    gym.spaces.Discrete)` in gym-side code holds, as it does for the reference (aqua.py:30-52) -- else the look-alikes of
    aquaticgymenv_amd/spaces.py (the build image has no gym)."""
    gym_module = _gym if gym_module is None else gym_module
    try:
        return gym_module.spaces.Box, gym_module.spaces.Discrete
    except Exception:
        return spaces.Box, spaces.Discrete


def make_space(cls, *args, seed=None, **kwargs):
    """cls(*args, **kwargs) seeded with `seed`: newer gym and the look-alikes take seed= in the constructor, gym 0.17-0.21
    (the reference's era) have a .seed() method instead"""
    try:
        return cls(*args, seed=seed, **kwargs)
    except TypeError:
        space = cls(*args, **kwargs)
        if seed is not None and hasattr(space, "seed"):
            space.seed(seed)
        return space


class _LazyInfo(dict):
    """info dict of a batched step: the three flags are produced from the term codes on first access."""

    def __init__(self, term):
        super().__init__()
        self["term"] = term

    def __missing__(self, key):
        if key in INFO_KEYS:
            value = self["term"] == (INFO_KEYS.index(key) + 1)
            self[key] = value
            return value
        raise KeyError(key)

    def __contains__(self, key):
        return key in INFO_KEYS or super().__contains__(key)


class AquaEnv(_EnvBase):
    metadata = {"render.modes": ["human"]}
    continuous = False

    def __init__(self, obstacles=False, waves=True, random_boat=True, random_goal=True, num_envs=1, device=None,
                 seed=None, autoreset="next_step"):
        self.has_waves = int(waves)
        self.random_boat = random_boat
        self.random_goal = random_goal
        self.num_envs = int(num_envs)
        # constants the reference publishes as attributes (aqua.py:19-25, 91, 94)
        self.world_size = 100
        self.motor_min_thrust = +0.2
        self.motor_max_thrust = +0.5
        self.wave_min_speed = -0.05 * self.has_waves
        self.wave_max_speed = +0.05 * self.has_waves
        self.wave_speed_variance = 0.001 * self.has_waves
        self.time_limit = TIME_LIMIT
        self.tau = 1
        Box, Discrete = space_types()
        if self.continuous:
            self.action_space = make_space(Box, self.motor_min_thrust, self.motor_max_thrust, shape=[2], dtype=np.float64,
                                           seed=seed)
        else:
            self.actions = [(0.2, 0.5), (0.5, 0.2), (0.5, 0.5)]      # left, right, straight (aqua.py:33-42)
            self.action_space = make_space(Discrete, len(self.actions), seed=seed)
        self.observation_space = make_space(Box, np.array([0, 0, -np.pi, 0, 0]),
                                            np.array([self.world_size, self.world_size, np.pi, self.world_size,
                                                      self.world_size]), dtype=np.float64, seed=seed)
        self.wave_space = make_space(Box, self.wave_min_speed, self.wave_max_speed, shape=[2], dtype=np.float64, seed=seed)
        rows = presets.rows_from(obstacles)
        self.obstacles = presets.as_reference_list(rows)
        self.core = BatchedAqua(self.num_envs, obstacles=rows, waves=waves, random_boat=random_boat,
                                random_goal=random_goal, continuous=self.continuous, device=device, seed=seed,
                                auto_reset=autoreset if self.num_envs > 1 else False)
        self._needs_reset = True

    # ------------------------------------------------------------------ reference-shaped API
    @property
    def time(self):
        t = self.core.time[: self.num_envs]
        return int(t[0]) if self.num_envs == 1 else t

    def seed(self, seed=None):
        if seed is not None:
            self.core.seed = int(seed)
        return [self.core.seed]

    def reset(self):
        obs = self.core.reset()
        self._needs_reset = False
        if self.num_envs == 1:
            return obs[0].to("cpu").numpy().astype(np.float64)
        return obs

    def step(self, action):
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        if self.num_envs == 1:
            return self._step_single(action)
        obs, reward, term = self.core.step(action)
        return obs, reward, term != 0, _LazyInfo(term)

    def _step_single(self, action):
        if self.continuous:
            a = np.asarray(action, dtype=np.float64).reshape(-1)
            if a.shape != (2,):
                raise ValueError("continuous action must have 2 entries")
            if not self.action_space.contains(a):
                print("input {0!r} provided is out of bounds and has been normalized".format(action))
            act = a.astype(np.float32).reshape(1, 2)
        else:
            idx = int(action)
            if isinstance(action, float) or idx != action:
                raise TypeError("list indices must be integers")
            if not -3 <= idx < 3:
                raise IndexError("list index out of range")      # what self.actions[action] raises (aqua.py:154)
            act = np.array([idx], dtype=np.int64)
        obs, reward, term = self.core.step(act)
        host = self.core.torch.cat([obs[0], reward[:1], term[:1].to(obs.dtype)]).to("cpu").numpy()
        code = int(host[6])
        info = {INFO_KEYS[0]: code == 1, INFO_KEYS[1]: code == 2, INFO_KEYS[2]: code == 3}
        if code == 0:
            rew = np.float64(host[5])
        else:
            rew = 10 if code == 3 else -10
        return host[0:5].astype(np.float64), rew, code != 0, info

    def render(self, mode="human"):
        raise NotImplementedError("rendering (aqua.py:215-371, a pyglet viewer) is outside the batched step() path")

    def close(self):
        pass

    # ------------------------------------------------------------------ batch extras
    def rollout(self, steps, actions=None, fused=False, keep_all=True):
        return self.core.rollout(steps, actions=actions, fused=fused, keep_all=keep_all)

    def normalize_angle(self, value):
        """aqua.py:128-133 (host helper; the device does this inside the step)."""
        lo, hi = self.observation_space.low[2], self.observation_space.high[2]
        width = hi - lo
        off = value - lo
        return (off - (math.floor(off / width) * width)) + lo


class AquaContinuousEnv(AquaEnv):
    continuous = True
