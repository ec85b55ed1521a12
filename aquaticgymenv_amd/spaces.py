"""Minimal stand-ins for the two gym space classes the reference exposes on its env
(gym_aqua/envs/aqua.py:30,43,46,52): callers read .shape/.low/.high/.n and call .sample()/.contains()
(main/impl/dqn.py:99-100, main/impl/utils.py:19-21, main/testing/test_random.py:20).
The facade uses them when classic gym is NOT importable (the build image; with gym present it uses gym's own classes,
gym_aqua/envs/aqua.py space_types()): they accept everything the reference's callers do with the real classes."""
import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float64, seed=None):
        if shape is not None:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        if self.low.shape != self.high.shape:
            raise ValueError("low and high differ in shape")
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.RandomState(seed)

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)


class Discrete(object):
    def __init__(self, n, seed=None):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.RandomState(seed)

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        return int(self._rng.randint(self.n))

    def contains(self, x):
        try:
            v = int(x)
        except (TypeError, ValueError):
            return False
        return v == x and 0 <= v < self.n

    __contains__ = contains

    def __repr__(self):
        return "Discrete(%d)" % self.n
