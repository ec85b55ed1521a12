"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (``aquaticgymenv_amd/``, ``gym_aqua/``) never does.

Two restatements of the reference's ``AquaEnv.step()`` (gym_aqua/envs/aqua.py:135-213 and the
helpers at :128-133, :373-439), both float64, both checked against the golden vectors produced
from the reference itself (tests/test_oracle_golden.py) -- parity PINNED for step();
"reset parity unpinned" (see aqua_oracle.c):

* ``COracle``      -- ctypes binding of ``oracle/libaqua_oracle.so`` (aqua_oracle.c): batched SoA,
                      Philox or injected noise, float32 reset specification, float32-state rollout.
* ``ScalarPort``   -- one env per Python object, numpy scalars/2-vectors, the same sequence of
                      operations as the reference's class; this is what bench.py times as the
                      ``cpu_baseline`` of kind "port" (the reference itself cannot travel to the
                      GPU box).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libaqua_oracle.so")
# AQUA_ORACLE_LIB: another build of the same source (oracle/Makefile `asan`), used as it is -- tests/test_sanitizers.py
_LIB_OVERRIDE = os.environ.get("AQUA_ORACLE_LIB")

ACTION_U8, ACTION_I32, ACTION_I64, ACTION_F32X2 = 0, 1, 2, 3


def build(force=False):
    if _LIB_OVERRIDE:
        return _LIB_OVERRIDE
    src = os.path.join(_HERE, "aqua_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libaqua_oracle.so"])
    return _LIB


def obstacle_rows(obstacles):
    """reference-style list [(np.array([x, y]), 'c', r) | (np.array([x, y]), 'r', (w, h))] or an
    [K][5] array (cx, cy, kind, a, b) -> float64 [K][5]."""
    if obstacles is None:
        return np.zeros((0, 5), dtype=np.float64)
    if isinstance(obstacles, np.ndarray):
        return np.ascontiguousarray(obstacles, dtype=np.float64).reshape(-1, 5)
    rows = []
    for pos, kind, dims in obstacles:
        if kind == "c":
            rows.append([float(pos[0]), float(pos[1]), 0.0, float(dims), 0.0])
        elif kind == "r":
            rows.append([float(pos[0]), float(pos[1]), 1.0, float(dims[0]), float(dims[1])])
        else:
            raise Exception("unknown obstacle type %r" % (kind,))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 5)


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


class COracle(object):
    def __init__(self):
        self.lib = ctypes.CDLL(build())
        L = self.lib
        c_d, c_f = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)
        c_i32, c_u8, c_i64 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int64)
        L.aqua_oracle_step.argtypes = [ctypes.c_int64, ctypes.c_int, c_d, ctypes.c_int, c_d, c_i32, ctypes.c_int,
                                       ctypes.c_void_p, c_d, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64,
                                       c_d, c_u8, c_d]
        L.aqua_oracle_step.restype = None
        L.aqua_oracle_reset.argtypes = [ctypes.c_int64, ctypes.c_int, c_d, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, c_u8, c_f, ctypes.c_int64,
                                        c_i32]
        L.aqua_oracle_reset.restype = None
        L.aqua_oracle_step_tables.argtypes = L.aqua_oracle_step.argtypes
        L.aqua_oracle_step_tables.restype = None
        L.aqua_oracle_reset_tables.argtypes = L.aqua_oracle_reset.argtypes
        L.aqua_oracle_reset_tables.restype = None
        L.aqua_oracle_rollout_f32.argtypes = [ctypes.c_int64, ctypes.c_int, c_d, ctypes.c_int, ctypes.c_int, c_f,
                                              ctypes.c_int64, c_i32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, c_f,
                                              c_u8, c_i64]
        L.aqua_oracle_rollout_f32.restype = ctypes.c_int64
        L.aqua_oracle_philox.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                         ctypes.POINTER(ctypes.c_uint32)]
        L.aqua_oracle_philox.restype = None
        L.aqua_oracle_step_noise.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, c_d,
                                             ctypes.POINTER(ctypes.c_uint32)]
        L.aqua_oracle_step_noise.restype = None
        L.aqua_oracle_threads.restype = ctypes.c_int

    def threads(self):
        return int(self.lib.aqua_oracle_threads())

    def philox(self, key, ctr):
        c = (ctypes.c_uint32 * 4)(*ctr)
        o = (ctypes.c_uint32 * 4)()
        self.lib.aqua_oracle_philox(key[0], key[1], c, o)
        return [int(v) for v in o]

    def step_noise(self, seed, env, tick):
        u = (ctypes.c_double * 2)()
        raw = (ctypes.c_uint32 * 4)()
        self.lib.aqua_oracle_step_noise(seed, env, tick, u, raw)
        return np.array([u[0], u[1]]), [int(v) for v in raw]

    @staticmethod
    def _action(action):
        a = np.ascontiguousarray(action)
        if a.dtype == np.uint8:
            return a, ACTION_U8
        if a.dtype == np.int32:
            return a, ACTION_I32
        if a.dtype == np.int64:
            return a, ACTION_I64
        if a.dtype == np.float32:
            return a, ACTION_F32X2        # SoA [2][n]
        raise TypeError("unsupported action dtype %s" % a.dtype)

    def step(self, state, time, action, obstacles=None, waves=1, noise_u=None, seed=0, tick=0, env_offset=0,
             want_margins=True):
        """state float64 [7][n] (x, y, theta, gx, gy, wx, wy), time int32 [n] -- both updated IN PLACE.
        action: uint8/int32/int64 [n] or float32 [2][n].  noise_u float64 [2][n] in [-1, 1) or None (Philox).
        Returns reward f64[n], term u8[n] (0 none, 1 collided, 2 time, 3 success), margins f64[3][n]."""
        assert state.dtype == np.float64 and state.flags.c_contiguous and state.shape[0] == 7
        assert time.dtype == np.int32 and time.flags.c_contiguous
        n = state.shape[1]
        obst = obstacle_rows(obstacles)
        act, kind = self._action(action)
        if kind == ACTION_F32X2:
            assert act.shape == (2, n)
        else:
            assert act.shape == (n,)
        if noise_u is not None:
            noise_u = np.ascontiguousarray(noise_u, dtype=np.float64)
            assert noise_u.shape == (2, n)
        reward = np.empty(n, dtype=np.float64)
        term = np.empty(n, dtype=np.uint8)
        margins = np.empty((3, n), dtype=np.float64) if want_margins else None
        self.lib.aqua_oracle_step(n, obst.shape[0], _p(obst, ctypes.c_double), int(waves), _p(state, ctypes.c_double),
                                  _p(time, ctypes.c_int32), kind, act.ctypes.data_as(ctypes.c_void_p),
                                  _p(noise_u, ctypes.c_double), int(seed), int(tick), int(env_offset),
                                  _p(reward, ctypes.c_double), _p(term, ctypes.c_uint8), _p(margins, ctypes.c_double))
        return reward, term, margins

    def reset(self, state, time, obstacles=None, waves=1, random_boat=True, random_goal=True, seed=0, tick=0,
              env_offset=0, mask=None):
        """state float32 [7][ld] (only the first n = len(time) columns are touched), IN PLACE."""
        assert state.dtype == np.float32 and state.flags.c_contiguous and state.shape[0] == 7
        n = time.shape[0]
        obst = obstacle_rows(obstacles)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mask.shape == (n,)
        self.lib.aqua_oracle_reset(n, obst.shape[0], _p(obst, ctypes.c_double), int(waves), int(random_boat),
                                   int(random_goal), int(seed), int(tick), int(env_offset), _p(mask, ctypes.c_uint8),
                                   _p(state, ctypes.c_float), state.shape[1], _p(time, ctypes.c_int32))

    def step_tables(self, state, time, action, tables, waves=1, noise_u=None, seed=0, tick=0, env_offset=0):
        """step() with one obstacle list per world: tables float64 [n][K][5] (kind < 0: absent row)."""
        assert state.dtype == np.float64 and state.flags.c_contiguous and state.shape[0] == 7
        n = state.shape[1]
        tables = np.ascontiguousarray(tables, dtype=np.float64)
        assert tables.ndim == 3 and tables.shape[0] == n and tables.shape[2] == 5
        act, kind = self._action(action)
        reward = np.empty(n, dtype=np.float64)
        term = np.empty(n, dtype=np.uint8)
        margins = np.empty((3, n), dtype=np.float64)
        if noise_u is not None:
            noise_u = np.ascontiguousarray(noise_u, dtype=np.float64)
        self.lib.aqua_oracle_step_tables(n, tables.shape[1], _p(tables, ctypes.c_double), int(waves),
                                         _p(state, ctypes.c_double), _p(time, ctypes.c_int32), kind,
                                         act.ctypes.data_as(ctypes.c_void_p), _p(noise_u, ctypes.c_double), int(seed),
                                         int(tick), int(env_offset), _p(reward, ctypes.c_double), _p(term, ctypes.c_uint8),
                                         _p(margins, ctypes.c_double))
        return reward, term, margins

    def reset_tables(self, state, time, tables, waves=1, random_boat=True, random_goal=True, seed=0, tick=0,
                     env_offset=0, mask=None):
        """reset() with one obstacle list per world (tables float64 [n][K][5])."""
        assert state.dtype == np.float32 and state.flags.c_contiguous and state.shape[0] == 7
        n = time.shape[0]
        tables = np.ascontiguousarray(tables, dtype=np.float64)
        assert tables.ndim == 3 and tables.shape[0] == n and tables.shape[2] == 5
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self.lib.aqua_oracle_reset_tables(n, tables.shape[1], _p(tables, ctypes.c_double), int(waves), int(random_boat),
                                          int(random_goal), int(seed), int(tick), int(env_offset),
                                          _p(mask, ctypes.c_uint8), _p(state, ctypes.c_float), state.shape[1],
                                          _p(time, ctypes.c_int32))

    def rollout_f32(self, state, time, steps, obstacles=None, waves=1, continuous=False, actions=None, seed=0,
                    tick0=0, env_offset=0, auto_reset=True):
        """float32-state rollout (state [7][ld], in place).  actions None -> sampled from Philox stream 4.
        auto_reset: False/0 none, True/1 same-step restart, 2 next-step restart (pending worlds: time == -1).
        Returns (episodes, reward f32[n] of the last step, term u8[n] of the last step, term_counts[3])."""
        assert state.dtype == np.float32 and state.flags.c_contiguous
        n = time.shape[0]
        obst = obstacle_rows(obstacles)
        reward = np.zeros(n, dtype=np.float32)
        term = np.zeros(n, dtype=np.uint8)
        counts = np.zeros(3, dtype=np.int64)
        kind, ptr = 0, None
        if actions is not None:
            actions, kind = self._action(actions)
            ptr = actions.ctypes.data_as(ctypes.c_void_p)
        ep = self.lib.aqua_oracle_rollout_f32(n, obst.shape[0], _p(obst, ctypes.c_double), int(waves), int(continuous),
                                              _p(state, ctypes.c_float), state.shape[1], _p(time, ctypes.c_int32),
                                              int(steps), ptr, kind, int(seed), int(tick0), int(env_offset),
                                              int(auto_reset), _p(reward, ctypes.c_float), _p(term, ctypes.c_uint8),
                                              _p(counts, ctypes.c_int64))
        return int(ep), reward, term, counts


# ----------------------------------------------------------------------------------------------
# reference-style scalar port (one env per object)
# ----------------------------------------------------------------------------------------------
class ScalarPort(object):
    """One world per object, float64, numpy on 2-vectors -- the cost profile of the reference's class
    (aqua.py:9-213).  Written from the semantics in SURVEY.md Appendix A.1; not a copy of the file.
    Noise comes from a private RandomState (the reference uses numpy's global one, aqua.py:188) or is
    injected through ``step(action, noise_u=...)``."""

    TERM_KEYS = ("Termination.collided", "Termination.time", "Termination.success")

    def __init__(self, obstacles=None, waves=True, continuous=False, seed=None):
        self.k_waves = int(waves)
        self.continuous = bool(continuous)
        self.obst = obstacle_rows(obstacles)
        self.lo = np.array([0.0, 0.0, -np.pi, 0.0, 0.0])
        self.hi = np.array([100.0, 100.0, np.pi, 100.0, 100.0])
        self.thrust_table = ((0.2, 0.5), (0.5, 0.2), (0.5, 0.5))       # aqua.py:33-42
        self.rng = np.random.RandomState(seed)
        self.boat = np.zeros(3)
        self.goal = np.zeros(2)
        self.wave = np.zeros(2)
        self.prev = None
        self.time = 0

    # geometry (aqua.py:373-390)
    @staticmethod
    def _gap_circles(p, pr, q, qr):
        return np.linalg.norm(p - q) - (pr + qr)

    @staticmethod
    def _gap_rect(center, dims, q, qr):
        half = np.array([dims[0] / 2, dims[1] / 2])
        nearest = np.clip(q, center - half, center + half)
        return np.linalg.norm(q - nearest) - qr

    def _hits_obstacle(self, p, pr):
        for row in self.obst:
            if row[2] == 0.0:
                gap = self._gap_circles(row[0:2], row[3], p, pr)
            else:
                gap = self._gap_rect(row[0:2], row[3:5], p, pr)
            if gap <= 0:
                return True
        return False

    def _hits_border(self, p, pr):
        return bool(np.any(p - pr < self.lo[0:2]) or np.any(p + pr > self.hi[0:2]))

    def _goal_gap(self, p):
        return self._gap_circles(self.goal, 2.5, p, 2.5)

    def set_state(self, boat, goal, wave, time):
        self.boat = np.array(boat, dtype=np.float64)
        self.goal = np.array(goal, dtype=np.float64)
        self.wave = np.array(wave, dtype=np.float64)
        self.time = int(time)

    def reset(self):
        """rejection sampling as aqua.py:100-126 (distribution only; RNG differs from gym's)."""
        while True:
            self.goal = self.rng.uniform(self.lo, self.hi)[3:]
            if not (self._hits_border(self.goal, 2.5) or self._hits_obstacle(self.goal, 2.5)):
                break
        while True:
            self.boat = self.rng.uniform(self.lo, self.hi)[:3]
            p = self.boat[0:2]
            if not (self._goal_gap(p) <= 0 or self._hits_border(p, 2.5) or self._hits_obstacle(p, 2.5)):
                break
        self.wave = self.rng.uniform(-0.05 * self.k_waves, 0.05 * self.k_waves, 2)
        self.time = 0
        return np.concatenate((self.boat, self.goal))

    def step(self, action, noise_u=None):
        if isinstance(action, np.ndarray):
            action = action.astype(np.float64)
        self.prev = self.boat.copy()
        self.time += 1
        if self.continuous:
            a = np.asarray(action, dtype=np.float64)
            if not (a.shape == (2,) and np.all(a >= 0.2) and np.all(a <= 0.5)):
                a = np.clip(a, 0.2, 0.5)
            v_left, v_right = a[0], a[1]
        else:
            v_left, v_right = self.thrust_table[action]
        gap = v_right - v_left
        gap = math.copysign(max(abs(gap), 1e-8), gap)
        radius = 2.5 / 2 * (v_right + v_left) / gap
        omega = gap / 2.5
        heading = np.pi / 2 + self.boat[2]
        pivot = self.boat[0:2] + radius * np.array([-np.sin(heading), np.cos(heading)])
        co, si = np.cos(omega), np.sin(omega)
        turn = np.array([[co, -si], [si, co]])
        self.boat[0:2] = turn.dot(self.boat[0:2] - pivot) + pivot + self.wave
        off = (self.boat[2] + omega) - self.lo[2]
        span = self.hi[2] - self.lo[2]
        self.boat[2] = (off - (math.floor(off / span) * span)) + self.lo[2]
        sigma = 0.001 * self.k_waves
        kick = self.rng.uniform(-sigma, sigma, 2) if noise_u is None else np.asarray(noise_u) * sigma
        self.wave = np.clip(self.wave + kick, -0.05 * self.k_waves, 0.05 * self.k_waves)
        info = dict.fromkeys(self.TERM_KEYS, False)
        done = True
        p = self.boat[0:2]
        if self._hits_obstacle(p, 2.5) or self._hits_border(p, 2.5):
            info[self.TERM_KEYS[0]] = True
            reward = -10
        elif self.time > 1000:
            info[self.TERM_KEYS[1]] = True
            reward = -10
        elif self._goal_gap(p) <= 0:
            info[self.TERM_KEYS[2]] = True
            reward = 10
        else:
            reward = (self._goal_gap(self.prev[0:2]) - self._goal_gap(p)) * 0.7
            done = False
        return np.concatenate((self.boat, self.goal)), reward, done, info


def time_scalar_port(obstacles, continuous, budget_s=10.0, seed=0):
    """random-action rollout with reset on done, for ~budget_s seconds; returns (steps, seconds)."""
    import time as _t
    env = ScalarPort(obstacles=obstacles, waves=True, continuous=continuous, seed=seed)
    env.reset()
    rng = np.random.RandomState(seed + 1)
    steps = 0
    t0 = _t.perf_counter()
    while True:
        for _ in range(2000):
            a = rng.uniform(0.2, 0.5, 2) if continuous else int(rng.randint(3))
            _, _, done, _ = env.step(a)
            if done:
                env.reset()
        steps += 2000
        dt = _t.perf_counter() - t0
        if dt >= budget_s:
            return steps, dt
