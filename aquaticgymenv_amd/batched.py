"""BatchedAqua: N independent AquaEnv worlds advanced in lock-step on one MI355X.

Host-side owner of the device buffers (PyTorch tensors used as raw HBM allocations) and caller of
the C ABI in include/aqua_hip.h.  There is no CPU path: constructing it without a HIP device or
without libaqua_hip.so raises.

Reference being replaced: one Python object per world, gym_aqua/envs/aqua.py:9-213.
"""
import contextlib
import ctypes

import numpy as np

from . import _capi, presets

TIME_LIMIT = 1000          # aqua.py:91
_NO_CONTEXT = contextlib.nullcontext()


def _round_up(n, m):
    return (n + m - 1) // m * m


class LaunchEvents(object):
    """Two HIP events for rollout(events=...): the first step launch's start and the last one's end, taken from the
    kernels' own dispatch packets (no marker packets in the queue).  elapsed_ms() after the stream has been synchronised."""

    def __init__(self):
        self._handles = []
        for _ in range(2):
            e = ctypes.c_void_p()
            rc = _capi.lib.aqua_event_create(ctypes.byref(e))
            if rc:
                self.close()
                _capi.check(rc, "aqua_event_create")
            self._handles.append(e)

    @property
    def start(self):
        return self._handles[0]

    @property
    def stop(self):
        return self._handles[1]

    def elapsed_ms(self):
        ms = ctypes.c_float(0.0)
        _capi.check(_capi.lib.aqua_event_elapsed_ms(self._handles[0], self._handles[1], ctypes.byref(ms)), "aqua_event_elapsed_ms")
        return float(ms.value)

    def close(self):
        for e in self._handles:
            _capi.lib.aqua_event_destroy(e)
        self._handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreparedRollout(object):
    """rollout(steps, ...) with its arguments checked and marshalled once (BatchedAqua.prepare_rollout): launch() is one C
    call -- what a captured graph is for replays, for launches that cannot be captured (events attached to them)."""

    def __init__(self, env, steps, args, tick_index, reward, term, keep):
        self._env, self.steps, self._args, self._tick_index = env, steps, list(args), tick_index
        self.reward, self.term = reward, term
        self._keep = keep                  # the tensors and events the marshalled pointers refer to
        self._clip_view = None             # continuous thrusts [T][2][N] whose clipped worlds every launch() counts

    def launch(self):
        env = self._env
        self._args[self._tick_index] = env._tick
        env._count_clipped(self._clip_view)            # the buffer as it is NOW: every launch steps with its current content
        _capi.check(_capi.lib.aqua_rollout_events_f32(*self._args), "aqua_rollout_events_f32")
        env._tick += self.steps
        return self.reward, self.term


class RolloutGraph(object):
    """A captured HIP graph of T batched steps (+ the device tick bump); replay with launch()."""

    def __init__(self, env, handle, steps, reward, term, events=None):
        self._env = env
        self._handle = handle
        self.steps = steps
        self.reward = reward
        self.term = term
        self._events = events             # (start, stop) AquaEvent handles recorded as nodes of the graph, or None
        self._clip_view = None            # continuous thrusts [T][2][N] whose clipped worlds every launch() counts
        self._launch = _capi.lib.aqua_graph_launch

    def launch(self, stream=None):
        """Replay the graph.  stream: the HIP stream as a ctypes.c_void_p marshalled once by a caller that replays in a tight
        loop (bench.py: ~3 us of host time per replay sit between an event recorded ahead of the launch and the launch
        itself, and on an idle stream they count: profiles/r04/launch_gap/); default: torch's current stream.  A stream other
        than torch's current one is fine: what launch() itself queues ahead of the replay (the tick-base refresh after a
        step()/reset()/restore() in between, the clipped-action count) is queued on THAT stream; the buffers the graph reads
        (actions, injected noise) are the caller's to order."""
        env = self._env
        if env._device_tick != env._tick or self._clip_view is not None:
            # the tick-base refresh and the clipped-action count are small torch operations: they must be ordered ahead of the
            # replay, so they go to the stream the replay goes to (torch's current one unless the caller named another)
            torch = env.torch
            other = stream is not None and stream.value != torch.cuda.current_stream(env.device).cuda_stream
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream.value, device=env.device)) if other else _NO_CONTEXT
            with ctx:
                if env._device_tick != env._tick:
                    env._sync_device_tick()
                if self._clip_view is not None:
                    env._count_clipped(self._clip_view)        # (count_clipped=True: a replay steps with the buffer's CURRENT content)
        rc = self._launch(self._handle, stream if stream is not None else env._stream())
        if rc:
            _capi.check(rc, "aqua_graph_launch")
        env._tick += self.steps
        env._device_tick += self.steps
        return self.reward, self.term

    def upload(self):
        """hipGraphUpload on the current stream: the first launch() then finds the graph's device-side resources in place"""
        _capi.check(_capi.lib.aqua_graph_upload(self._handle, self._env._stream()), "aqua_graph_upload")

    def elapsed_ms(self):
        """GPU time of the last replay between the graph's first and last node (capture_rollout(timing=True)); the
        stream must have been synchronised since."""
        if self._events is None:
            raise RuntimeError("captured without timing=True")
        ms = ctypes.c_float(0.0)
        _capi.check(_capi.lib.aqua_event_elapsed_ms(self._events[0], self._events[1], ctypes.byref(ms)), "aqua_event_elapsed_ms")
        return float(ms.value)

    def close(self):
        if self._handle is not None:
            _capi.lib.aqua_graph_destroy(self._handle)
            self._handle = None
        if self._events is not None:
            for e in self._events:
                _capi.lib.aqua_event_destroy(e)
            self._events = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchedAqua(object):
    """N worlds in struct-of-arrays float32 buffers on one GPU.

    state   float32 [7][ld]   x, y, theta, goal_x, goal_y, wave_x, wave_y   (rows 0..4 are the observation)
    time    int32   [ld]
    reward  float32 [ld],  term uint8 [ld] (0 none, 1 collided, 2 time, 3 success)
    done_bits int64 [ceil(N/64)] (bit-packed done mask straight from the wavefront ballots)
    """

    def __init__(self, num_envs, obstacles=False, waves=True, random_boat=True, random_goal=True, continuous=False,
                 device=None, seed=None, env_offset=0, auto_reset=True, normalized_obs=False, count_clipped=False):
        import torch
        self.torch = torch
        if num_envs < 1:
            raise ValueError("num_envs must be >= 1")
        dev = torch.device("cuda" if device is None else device)
        if dev.type != "cuda":
            raise RuntimeError("BatchedAqua runs on an AMD GPU through HIP only (device=%r); there is no CPU path" % (device,))
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: BatchedAqua has no CPU path")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.num_envs = int(num_envs)
        self.env_offset = int(env_offset)
        self.continuous = bool(continuous)
        # restart of finished worlds: False/0 never (the reference's behaviour), True/1/"same_step" inside the
        # launch that finished them, 2/"next_step" during the next step (Gymnasium >= 1.0 convention, fastest)
        modes = {False: 0, True: 1, 0: 0, 1: 1, 2: 2, "none": 0, "same_step": 1, "next_step": 2}
        if auto_reset not in modes:
            raise ValueError("auto_reset must be False, True, 'same_step' or 'next_step'")
        self.auto_reset = modes[auto_reset]
        self.has_waves = int(waves)                   # aqua.py:15
        self.seed = int(seed) if seed is not None else int(np.random.SeedSequence().entropy & ((1 << 64) - 1))
        # obstacles: False/True/list/[K][5] -> ONE list for the whole batch (presets.rows_from);
        #            float [N][K][5] -> one list PER WORLD (rows with kind < 0 are absent), include/aqua_hip.h
        self.per_world = isinstance(obstacles, np.ndarray) and obstacles.ndim == 3
        if self.per_world:
            if obstacles.shape[0] != num_envs or obstacles.shape[2] != 5 or obstacles.shape[1] < 1:
                raise ValueError("per-world obstacle tables must have shape [num_envs][K >= 1][5]")
            self.obstacle_tables = np.ascontiguousarray(obstacles, dtype=np.float64)
            self.obstacle_rows = presets.rows_from(False)
            self.K = int(obstacles.shape[1])
        else:
            self.obstacle_rows = presets.rows_from(obstacles)
            self.K = int(self.obstacle_rows.shape[0])
        self.params = _capi.AquaParams(waves=self.has_waves, continuous=int(self.continuous),
                                       random_boat=int(bool(random_boat)), random_goal=int(bool(random_goal)),
                                       time_limit=TIME_LIMIT)
        n = self.num_envs
        self.ld = _round_up(n, 64)
        with torch.cuda.device(dev):
            self.state = torch.zeros((7, self.ld), dtype=torch.float32, device=dev)
            self.time = torch.zeros(self.ld, dtype=torch.int32, device=dev)
            self.reward = torch.zeros(self.ld, dtype=torch.float32, device=dev)
            self.term = torch.zeros(self.ld, dtype=torch.uint8, device=dev)
            self.done_bits = torch.zeros(self.ld // 64, dtype=torch.int64, device=dev)
            self._tab32 = self._tab64 = None
            self._r_max = 0.0
            if self.per_world:
                # [K][6][ld] struct of arrays + the world-major copy [ld][K][6] behind it (include/aqua_hip.h)
                t32 = np.zeros(int(_capi.lib.aqua_tables32_floats(self.K, self.ld)), dtype=np.float32)
                t64 = np.zeros((self.K, 5, self.ld), dtype=np.float64)
                r_max = ctypes.c_float(0.0)
                _capi.check(_capi.lib.aqua_pack_tables(self.obstacle_tables.ctypes.data, self.K, n, self.ld, t32.ctypes.data,
                                                       t64.ctypes.data, ctypes.byref(r_max)), "aqua_pack_tables")
                self._tab32, self._tab64, self._r_max = torch.from_numpy(t32).to(dev), torch.from_numpy(t64).to(dev), r_max.value
            blob = _capi.pack_obstacles(self.obstacle_rows) if not self.per_world else b""
            if blob:
                host = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
                self._blob = host.to(dev)
            else:
                self._blob = None
            self._tick_dev = torch.zeros(2, dtype=torch.int64, device=dev)      # [0] tick base of captured graphs, [1] scratch
            # optional fused epilogue (main/impl/utils.py:15-33): obs / (high - low), angle + 0.5
            self.obs_norm_buf = torch.zeros((5, self.ld), dtype=torch.float32, device=dev) if normalized_obs else None
        # aqua.py:145-150 prints a message when a continuous action is outside [0.2, 0.5] and clips it.  The kernels clip
        # silently; with count_clipped=True every step()/rollout() with a caller-provided action buffer also adds the number of
        # WORLDS whose action was clipped to this device counter (two small torch reductions, no sync)
        self.clipped_actions = (torch.zeros((), dtype=torch.int64, device=dev) if (count_clipped and self.continuous) else None)
        self._tick = 0                    # one per step: draws of the step, and of the restarts the step kernels do
        self._device_tick = 0
        self._resets = 0                  # one per reset() call: its draws use tick RESET_TICK_BASE + _resets, so
                                          # that a reset between two steps does not shift the steps' tick parity
                                          # (the next-step restart markers carry it, include/aqua_hip.h)
        self._action_soa = None           # staging for (N, 2) -> [2][ld] continuous actions

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _blob_ptr(self):
        return self._blob.data_ptr() if self._blob is not None else None

    def _sync_device_tick(self):
        if self._device_tick != self._tick:
            self._tick_dev[0:1].fill_(self._tick)
            self._device_tick = self._tick

    @property
    def obs(self):
        """[N, 5] view (x, y, theta, goal_x, goal_y) of the state rows -- no copy (aqua.py:213)."""
        return self.state[:5, :self.num_envs].t()

    @property
    def obs_norm(self):
        """[N, 5] view of the normalised observation written by the step kernels (normalized_obs=True)."""
        if self.obs_norm_buf is None:
            raise RuntimeError("construct with normalized_obs=True")
        return self.obs_norm_buf[:, :self.num_envs].t()

    def _norm_ptr(self):
        return self.obs_norm_buf.data_ptr() if self.obs_norm_buf is not None else None

    @property
    def wave(self):
        return self.state[5:7, :self.num_envs].t()

    def _as_action(self, action, soa=False):
        """-> (tensor kept alive, pointer, kind, action_ld)"""
        torch = self.torch
        n = self.num_envs
        if self.continuous:
            a = action
            if not isinstance(a, torch.Tensor):
                a = torch.as_tensor(np.asarray(a, dtype=np.float32))
            a = a.to(device=self.device, dtype=torch.float32)
            if soa:
                if a.dim() != 2 or a.shape[0] != 2 or a.shape[1] < n or a.stride(1) != 1:
                    raise ValueError("soa action must be a float32 [2][>=N] tensor with unit inner stride")
                self._count_clipped(a[:, :n])
                return a, a.data_ptr(), _capi.ACT_F32X2, a.stride(0)
            if a.dim() == 1 and n == 1:
                a = a.reshape(1, 2)
            if a.shape != (n, 2):
                raise ValueError("continuous action must have shape (%d, 2), got %s" % (n, tuple(a.shape)))
            if self._action_soa is None:
                self._action_soa = torch.empty((2, self.ld), dtype=torch.float32, device=self.device)
            self._action_soa[:, :n].copy_(a.t())
            self._count_clipped(self._action_soa[:, :n])
            return self._action_soa, self._action_soa.data_ptr(), _capi.ACT_F32X2, self.ld
        a = action
        if not isinstance(a, torch.Tensor):
            arr = np.asarray(a)
            if arr.dtype.kind not in "iu":
                raise TypeError("discrete actions must be integers, got dtype %s" % arr.dtype)
            a = torch.as_tensor(arr.astype(np.int64).reshape(-1))
        if a.dtype == torch.uint8:
            kind = _capi.ACT_U8
        elif a.dtype == torch.int32:
            kind = _capi.ACT_I32
        elif a.dtype == torch.int64:
            kind = _capi.ACT_I64
        else:
            raise TypeError("discrete actions must be uint8 / int32 / int64, got %s" % a.dtype)
        a = a.to(self.device).reshape(-1)
        if a.numel() != n or not a.is_contiguous():
            if a.numel() != n:
                raise ValueError("expected %d actions, got %d" % (n, a.numel()))
            a = a.contiguous()
        return a, a.data_ptr(), kind, 0

    def _count_clipped(self, thrusts):
        """thrusts float32 [..., 2, n]: adds the worlds with a thrust outside [0.2, 0.5] (aqua.py:145-150) to the counter"""
        if self.clipped_actions is not None and thrusts is not None:
            out = (thrusts < 0.2) | (thrusts > 0.5)
            self.clipped_actions += out.any(dim=-2).sum()

    # ------------------------------------------------------------------ the path
    RESET_TICK_BASE = 1 << 40             # reset() draws live far above any step tick

    def reset(self, mask=None):
        """(Re)start worlds: all of them, or those with mask[i] != 0 (uint8/bool tensor).  aqua.py:100-126."""
        torch = self.torch
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device)
            if mask.dtype == torch.bool:
                mask = mask.to(torch.uint8)
            if mask.dtype != torch.uint8 or mask.numel() < self.num_envs:
                raise ValueError("mask must be a uint8/bool tensor with one entry per world")
            mask = mask.contiguous()
            mptr = mask.data_ptr()
        with torch.cuda.device(self.device):
            if self.per_world:
                _capi.check(_capi.lib.aqua_reset_tables_f32(ctypes.byref(self.params), self._tab32.data_ptr(), self.K, self.ld,
                                                            self.num_envs, self.env_offset, self.state.data_ptr(), self.ld,
                                                            self.time.data_ptr(), mptr, self.seed,
                                                            self.RESET_TICK_BASE + self._resets, None, self._stream()),
                            "aqua_reset_tables_f32")
            else:
                _capi.check(_capi.lib.aqua_reset_f32(ctypes.byref(self.params), self._blob_ptr(), self.K, self.num_envs,
                                                     self.env_offset, self.state.data_ptr(), self.ld, self.time.data_ptr(),
                                                     mptr, self.seed, self.RESET_TICK_BASE + self._resets, None,
                                                     self._stream()), "aqua_reset_f32")
            if self._norm_ptr() is not None:          # the step kernels' epilogue, for the worlds just placed
                _capi.check(_capi.lib.aqua_obs_norm_f32(self.state.data_ptr(), self.ld, self.num_envs, mptr,
                                                        self._norm_ptr(), self._stream()), "aqua_obs_norm_f32")
        self._resets += 1
        return self.obs

    def _device_policy(self, name):
        if name in (True, "random"):
            return _capi.ACT_SAMPLE_C if self.continuous else _capi.ACT_SAMPLE_D
        if name == "bearing":
            if self.continuous:
                raise ValueError("the bearing policy (main/testing/test_optimal.py) is defined for discrete actions")
            return _capi.ACT_BEARING
        raise ValueError("unknown on-device policy %r (use 'random' or 'bearing')" % (name,))

    def step(self, action=None, soa=False, noise=None, sample_actions=False, policy=None):
        """One batched step (aqua.py:135-213).  Returns (obs view [N,5], reward [N], term [N] uint8).
        noise: optional float32 [2][>=N] uniforms in [-1, 1) replacing the Philox draws (parity tests).
        policy: 'random' (== sample_actions) or 'bearing': the action is produced on the device."""
        torch = self.torch
        n = self.num_envs
        if sample_actions or policy is not None:
            keep, aptr, kind, ald = None, None, self._device_policy(policy or "random"), 0
        else:
            keep, aptr, kind, ald = self._as_action(action, soa)
        nptr, nld = None, 0
        if noise is not None:
            if noise.dtype != torch.float32 or noise.dim() != 2 or noise.shape[0] != 2 or noise.shape[1] < n \
                    or noise.stride(1) != 1:
                raise ValueError("noise must be float32 [2][>=N] with unit inner stride")
            nptr, nld = noise.data_ptr(), noise.stride(0)
        with torch.cuda.device(self.device):
            if self.per_world:
                _capi.check(_capi.lib.aqua_step_tables_f32(ctypes.byref(self.params), self._tab32.data_ptr(),
                                                           self._tab64.data_ptr(), self.K, self.ld, self._r_max, n,
                                                           self.env_offset, self.state.data_ptr(), self.ld,
                                                           self.time.data_ptr(), aptr, kind, ald, nptr, nld, self.seed,
                                                           self._tick, None, self.reward.data_ptr(), self.term.data_ptr(),
                                                           self.done_bits.data_ptr(), self._norm_ptr(), int(self.auto_reset),
                                                           self._stream()),
                            "aqua_step_tables_f32")           # (same-step restart happens inside the launch)
            else:
                _capi.check(_capi.lib.aqua_step_f32(ctypes.byref(self.params), self._blob_ptr(), self.K, n, self.env_offset,
                                                    self.state.data_ptr(), self.ld, self.time.data_ptr(), aptr, kind, ald,
                                                    nptr, nld, self.seed, self._tick, None, self.reward.data_ptr(),
                                                    self.term.data_ptr(), self.done_bits.data_ptr(), self._norm_ptr(),
                                                    int(self.auto_reset), self._stream()), "aqua_step_f32")
        self._tick += 1
        del keep
        return self.obs, self.reward[:n], self.term[:n]

    def capture_step(self, action, noise=None, soa=False):
        """step(action, noise=noise) captured into a HIP graph: every launch() of the returned RolloutGraph is one batched
        step that reads `action` (and `noise`) from the caller's device buffers AS THEY ARE THEN -- the loop of
        main/testing/__init__.py:25-34 with the policy writing its actions into a fixed buffer between replays, no
        per-step marshalling.  action: discrete uint8/int32/int64 [N] or, continuous, float32 [2][>=N] with soa=True, on
        this device (a buffer step() would have to convert or stage cannot be re-read by a graph: ValueError)."""
        torch = self.torch
        if not isinstance(action, torch.Tensor) or action.device != self.device:
            raise ValueError("capture_step(): the action buffer must be a tensor on %s" % (self.device,))
        if self.continuous and not soa:
            raise ValueError("capture_step(): continuous actions as float32 [2][>=N] with soa=True")
        counter, self.clipped_actions = self.clipped_actions, None      # (nothing steps at capture time)
        try:
            keep, aptr, kind, ald = self._as_action(action, soa)
        finally:
            self.clipped_actions = counter
        if aptr != action.data_ptr():
            raise ValueError("capture_step(): the action buffer must be usable as it is (dtype, contiguity)")
        nptr, nld = None, 0
        if noise is not None:
            if noise.dtype != torch.float32 or noise.dim() != 2 or noise.shape[0] != 2 or noise.shape[1] < self.num_envs \
                    or noise.stride(1) != 1 or noise.device != self.device:
                raise ValueError("noise must be float32 [2][>=N] with unit inner stride on %s" % (self.device,))
            nptr, nld = noise.data_ptr(), noise.stride(0)
        lib, n = _capi.lib, self.num_envs
        self._sync_device_tick()
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        handle = ctypes.c_void_p()
        try:
            with torch.cuda.device(self.device), torch.cuda.stream(cap):
                s = self._stream()
                _capi.check(lib.aqua_graph_begin(s), "aqua_graph_begin")
                try:
                    tb = self._tick_dev.data_ptr()
                    if self.per_world:
                        rc = lib.aqua_step_tables_f32(ctypes.byref(self.params), self._tab32.data_ptr(), self._tab64.data_ptr(),
                                                      self.K, self.ld, self._r_max, n, self.env_offset, self.state.data_ptr(),
                                                      self.ld, self.time.data_ptr(), aptr, kind, ald, nptr, nld, self.seed, 0,
                                                      tb, self.reward.data_ptr(), self.term.data_ptr(),
                                                      self.done_bits.data_ptr(), self._norm_ptr(), int(self.auto_reset), s)
                    else:
                        rc = lib.aqua_step_f32(ctypes.byref(self.params), self._blob_ptr(), self.K, n, self.env_offset,
                                               self.state.data_ptr(), self.ld, self.time.data_ptr(), aptr, kind, ald, nptr, nld,
                                               self.seed, 0, tb, self.reward.data_ptr(), self.term.data_ptr(),
                                               self.done_bits.data_ptr(), self._norm_ptr(), int(self.auto_reset), s)
                    if rc == 0:
                        rc = lib.aqua_tick_advance(tb, 1, s)
                finally:
                    rc_end = lib.aqua_graph_end(s, ctypes.byref(handle))
                if rc != 0 and rc_end == 0:
                    lib.aqua_graph_destroy(handle)
                _capi.check(rc, "capture step")
                _capi.check(rc_end, "aqua_graph_end")
        finally:
            torch.cuda.current_stream(self.device).wait_stream(cap)
        g = RolloutGraph(self, handle, 1, self.reward, self.term)
        g._actions = (keep, noise)
        if self.clipped_actions is not None:
            g._clip_view = action[:, :n]
        g.done_history = None
        return g

    def _clip_view(self, steps, actions):
        """the thrusts a rollout of `steps` steps reads from `actions`, when there is a clipped-action counter to feed"""
        if self.clipped_actions is None or actions is None or isinstance(actions, str):
            return None
        return actions[:steps, :, :self.num_envs]

    def _rollout_args(self, steps, actions, soa_ld, count=True):
        """count=False: the caller (a captured graph, a prepared rollout) counts clipped actions at every launch() instead"""
        torch = self.torch
        n = self.num_envs
        if actions is None or isinstance(actions, str):
            return None, self._device_policy(actions or "random"), 0, 0
        if self.continuous:
            if actions.dtype != torch.float32 or actions.dim() != 3 or actions.shape[0] < steps or actions.shape[1] != 2 \
                    or actions.shape[2] < n or actions.stride(2) != 1:
                raise ValueError("continuous rollout actions must be float32 [T][2][>=N]")
            if count:
                self._count_clipped(actions[:steps, :, :n])
            return actions.data_ptr(), _capi.ACT_F32X2, actions.stride(1), actions.stride(0)
        kinds = {torch.uint8: _capi.ACT_U8, torch.int32: _capi.ACT_I32, torch.int64: _capi.ACT_I64}
        if actions.dtype not in kinds or actions.dim() != 2 or actions.shape[0] < steps or actions.shape[1] < n \
                or actions.stride(1) != 1:
            raise ValueError("discrete rollout actions must be uint8/int32/int64 [T][>=N]")
        return actions.data_ptr(), kinds[actions.dtype], 0, actions.stride(0)

    def _rollout_out(self, steps, keep_all):
        torch = self.torch
        if keep_all:
            reward = torch.empty((steps, self.ld), dtype=torch.float32, device=self.device)
            term = torch.empty((steps, self.ld), dtype=torch.uint8, device=self.device)
            return reward, term, self.ld
        return self.reward, self.term, 0

    def _done_out(self, steps, done_history):
        """done_history: None -> every step overwrites self.done_bits; True -> a fresh int64 [T][ld/64]
        tensor; or a caller-provided tensor of that shape (e.g. one half of a double buffer)."""
        torch = self.torch
        if done_history is None or done_history is False:
            return self.done_bits, 0
        if done_history is True:
            done_history = torch.zeros((steps, self.ld // 64), dtype=torch.int64, device=self.device)
        if done_history.dtype != torch.int64 or done_history.dim() != 2 or done_history.shape[0] < steps \
                or done_history.shape[1] < self.ld // 64 or done_history.stride(1) != 1:
            raise ValueError("done_history must be int64 [T][>= ld/64]")
        return done_history, done_history.stride(0)

    def rollout(self, steps, actions=None, fused=False, keep_all=True, done_history=None, events=None):
        """`steps` consecutive batched steps queued from C without returning to Python.
        events: a LaunchEvents (or a (start, stop) pair of which either may be None) stamped by the first launch's start and
                the last launch's end (one launch per step, one obstacle table for the batch only).
        actions: None / 'random' -> uniform random actions sampled on the device; 'bearing' -> the bearing policy
                 of main/testing/test_optimal.py evaluated on the device;
                 discrete: uint8/int32/int64 [T][>=N]; continuous: float32 [T][2][>=N].
        fused=True runs them as ONE launch with the state held in registers.
        Returns (reward, term): [T][ld] tensors when keep_all, else the last step's [ld] buffers."""
        torch = self.torch
        aptr, kind, ald, astride = self._rollout_args(steps, actions, self.ld)
        reward, term, ostride = self._rollout_out(steps, keep_all)
        done, dstride = self._done_out(steps, done_history)
        lib = _capi.lib
        if events is not None and (fused or self.per_world):
            raise ValueError("rollout(events=...) times one launch per step on a batch with one obstacle table")
        with torch.cuda.device(self.device):
            if self.per_world and fused:
                _capi.check(self._rollout_tables_fused(steps, aptr, kind, ald, astride, self._tick, None, reward, term, ostride,
                                                       self._stream()), "aqua_rollout_tables_fused_f32")
            elif self.per_world:
                _capi.check(self._rollout_tables(steps, aptr, kind, ald, astride, self._tick, None, reward, term, ostride,
                                                 done, dstride, 0, self._stream()), "aqua_rollout_tables_f32")
            elif fused:
                _capi.check(lib.aqua_rollout_fused_f32(ctypes.byref(self.params), self._blob_ptr(), self.K, self.num_envs,
                                                       self.env_offset, self.state.data_ptr(), self.ld,
                                                       self.time.data_ptr(), steps, aptr, kind, ald, astride, self.seed,
                                                       self._tick, None, reward.data_ptr(), term.data_ptr(), ostride,
                                                       int(self.auto_reset), self._stream()), "aqua_rollout_fused_f32")
            else:
                ev0, ev1 = (events.start, events.stop) if isinstance(events, LaunchEvents) else (events or (None, None))
                _capi.check(lib.aqua_rollout_events_f32(ctypes.byref(self.params), self._blob_ptr(), self.K, self.num_envs,
                                                        self.env_offset, self.state.data_ptr(), self.ld, self.time.data_ptr(),
                                                        steps, aptr, kind, ald, astride, self.seed, self._tick, None,
                                                        reward.data_ptr(), term.data_ptr(), ostride, done.data_ptr(),
                                                        dstride, self._norm_ptr(), int(self.auto_reset), 0, ev0, ev1,
                                                        self._stream()),
                            "aqua_rollout_events_f32")
        self._tick += steps
        return reward, term

    def prepare_rollout(self, steps, actions=None, keep_all=False, done_history=None, events=None):
        """rollout(steps, actions, keep_all=keep_all, done_history=done_history, events=events) as an object whose launch()
        is a single C call (one launch per step, one obstacle table for the batch; queued on the stream current NOW)."""
        if self.per_world:
            raise ValueError("prepare_rollout(): batches with one obstacle table")
        aptr, kind, ald, astride = self._rollout_args(steps, actions, self.ld, count=False)
        reward, term, ostride = self._rollout_out(steps, keep_all)
        done, dstride = self._done_out(steps, done_history)
        ev0, ev1 = (events.start, events.stop) if isinstance(events, LaunchEvents) else (events or (None, None))
        with self.torch.cuda.device(self.device):
            stream = self._stream()
        args = [ctypes.byref(self.params), self._blob_ptr(), self.K, self.num_envs, self.env_offset, self.state.data_ptr(),
                self.ld, self.time.data_ptr(), steps, aptr, kind, ald, astride, self.seed, self._tick, None,
                reward.data_ptr(), term.data_ptr(), ostride, done.data_ptr(), dstride, self._norm_ptr(),
                int(self.auto_reset), 0, ev0, ev1, stream]
        p = PreparedRollout(self, steps, args, 14, reward, term, (actions, done, events, reward, term))
        p._clip_view = self._clip_view(steps, actions)
        return p

    def _rollout_tables(self, steps, aptr, kind, ald, astride, tick, tick_base, reward, term, ostride, done, dstride, advance, s):
        return _capi.lib.aqua_rollout_tables_f32(ctypes.byref(self.params), self._tab32.data_ptr(), self._tab64.data_ptr(),
                                                 self.K, self.ld, self._r_max, self.num_envs, self.env_offset,
                                                 self.state.data_ptr(), self.ld, self.time.data_ptr(), steps, aptr, kind, ald,
                                                 astride, self.seed, tick, tick_base, reward.data_ptr(), term.data_ptr(),
                                                 ostride, done.data_ptr(), dstride, self._norm_ptr(), int(self.auto_reset),
                                                 advance, s)

    def _rollout_tables_fused(self, steps, aptr, kind, ald, astride, tick, tick_base, reward, term, ostride, s):
        return _capi.lib.aqua_rollout_tables_fused_f32(ctypes.byref(self.params), self._tab32.data_ptr(), self._tab64.data_ptr(),
                                                       self.K, self.ld, self._r_max, self.num_envs, self.env_offset,
                                                       self.state.data_ptr(), self.ld, self.time.data_ptr(), steps, aptr, kind,
                                                       ald, astride, self.seed, tick, tick_base, reward.data_ptr(),
                                                       term.data_ptr(), ostride, int(self.auto_reset), s)

    def capture_rollout(self, steps, actions=None, fused=False, keep_all=False, done_history=None, timing=False):
        """Capture `steps` batched steps into a HIP graph.  Noise stays fresh across replays: the
        kernels add a device-resident tick base that the graph's last node advances by `steps`.
        timing=True: the graph starts and ends with an event-record node (RolloutGraph.elapsed_ms())."""
        torch = self.torch
        aptr, kind, ald, astride = self._rollout_args(steps, actions, self.ld, count=False)     # (nothing steps at capture time)
        reward, term, ostride = self._rollout_out(steps, keep_all)
        done, dstride = self._done_out(steps, done_history)
        lib = _capi.lib
        self._sync_device_tick()
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        handle = ctypes.c_void_p()
        events = None
        if timing:
            events = (ctypes.c_void_p(), ctypes.c_void_p())
            for e in events:
                rc_ev = lib.aqua_event_create(ctypes.byref(e))
                if rc_ev != 0:                    # the first event (if it exists) must not outlive the failure
                    for made in events:
                        lib.aqua_event_destroy(made)
                    _capi.check(rc_ev, "aqua_event_create")
        try:
            return self._capture(cap, handle, events, steps, actions, fused, aptr, kind, ald, astride, reward, term, ostride,
                                 done, dstride, timing)
        except Exception:
            # a capture that failed (e.g. event-record nodes unsupported: bench.StepRunner.prepare falls back) leaves no
            # event handles behind, and the current stream is joined with the capture stream either way
            if events is not None:
                for e in events:
                    lib.aqua_event_destroy(e)
            raise
        finally:
            torch.cuda.current_stream(self.device).wait_stream(cap)

    def _capture(self, cap, handle, events, steps, actions, fused, aptr, kind, ald, astride, reward, term, ostride, done, dstride,
                 timing):
        torch, lib = self.torch, _capi.lib
        with torch.cuda.device(self.device), torch.cuda.stream(cap):
            s = self._stream()
            _capi.check(lib.aqua_graph_begin(s), "aqua_graph_begin")
            try:
                tb = self._tick_dev.data_ptr()
                if self.per_world and fused:
                    rc = self._rollout_tables_fused(steps, aptr, kind, ald, astride, 0, tb, reward, term, ostride, s)
                elif self.per_world:
                    rc = self._rollout_tables(steps, aptr, kind, ald, astride, 0, tb, reward, term, ostride, done, dstride, 1, s)
                elif fused:
                    rc = lib.aqua_rollout_fused_f32(ctypes.byref(self.params), self._blob_ptr(), self.K, self.num_envs,
                                                    self.env_offset, self.state.data_ptr(), self.ld,
                                                    self.time.data_ptr(), steps, aptr, kind, ald, astride, self.seed, 0,
                                                    tb, reward.data_ptr(), term.data_ptr(), ostride,
                                                    int(self.auto_reset), s)
                else:
                    rc = lib.aqua_rollout_f32(ctypes.byref(self.params), self._blob_ptr(), self.K, self.num_envs,
                                              self.env_offset, self.state.data_ptr(), self.ld, self.time.data_ptr(),
                                              steps, aptr, kind, ald, astride, self.seed, 0, tb, reward.data_ptr(),
                                              term.data_ptr(), ostride, done.data_ptr(), dstride, self._norm_ptr(),
                                              int(self.auto_reset), 1, s)        # advances the tick base itself
                if rc == 0 and fused:
                    rc = lib.aqua_tick_advance(tb, steps, s)
            finally:
                if timing:
                    rc_end = lib.aqua_graph_end_timed(s, ctypes.byref(handle), events[0], events[1])
                else:
                    rc_end = lib.aqua_graph_end(s, ctypes.byref(handle))
            if rc != 0 and rc_end == 0:           # the launches failed but a graph was made: do not keep it
                lib.aqua_graph_destroy(handle)
            _capi.check(rc, "capture rollout")
            _capi.check(rc_end, "aqua_graph_end")
        g = RolloutGraph(self, handle, steps, reward, term, events)
        g._actions = actions          # keep the action buffer alive as long as the graph
        g._clip_view = self._clip_view(steps, actions)
        g.done_history = done if dstride else None
        return g

    # ------------------------------------------------------------------ checkpoint (checkpoint / resume)
    def state_dict(self):
        """Everything a later load_state_dict() needs to continue this batch bit for bit: the worlds' state and time markers
        (incl. the next-step restart bookkeeping), the last step's outputs, the step / reset counters that key the Philox
        draws, seed, env_offset, restart mode and the obstacle table(s).  Host (CPU) tensors and plain Python values."""
        torch = self.torch
        torch.cuda.synchronize(self.device)
        d = {
            "format": 1, "num_envs": self.num_envs, "env_offset": self.env_offset, "seed": self.seed,
            "continuous": self.continuous, "auto_reset": int(self.auto_reset), "waves": self.has_waves,
            "random_boat": int(self.params.random_boat), "random_goal": int(self.params.random_goal),
            "tick": self._tick, "resets": self._resets, "per_world": self.per_world,
            "obstacle_rows": np.array(self.obstacle_rows, dtype=np.float64, copy=True),
            "state": self.state.cpu().clone(), "time": self.time.cpu().clone(), "reward": self.reward.cpu().clone(),
            "term": self.term.cpu().clone(), "done_bits": self.done_bits.cpu().clone(),
        }
        if self.per_world:
            d["obstacle_tables"] = np.array(self.obstacle_tables, copy=True)
        if self.obs_norm_buf is not None:
            d["obs_norm"] = self.obs_norm_buf.cpu().clone()
        if self.clipped_actions is not None:
            d["clipped_actions"] = int(self.clipped_actions)
        return d

    def load_state_dict(self, d):
        """Continue where state_dict() stopped.  The batch must have been constructed with the same shape, seed, offsets,
        restart mode and obstacles (checked): a checkpoint restores a run, it does not reconfigure one."""
        same = [("num_envs", self.num_envs), ("env_offset", self.env_offset), ("seed", self.seed), ("continuous", self.continuous),
                ("auto_reset", int(self.auto_reset)), ("waves", self.has_waves), ("per_world", self.per_world),
                ("random_boat", int(self.params.random_boat)), ("random_goal", int(self.params.random_goal))]
        if d.get("format") != 1:
            raise ValueError("unknown checkpoint format %r" % (d.get("format"),))
        for key, mine in same:
            if d[key] != mine:
                raise ValueError("checkpoint has %s=%r, this batch %r" % (key, d[key], mine))
        tables = d["obstacle_tables"] if self.per_world else d["obstacle_rows"]
        mine = self.obstacle_tables if self.per_world else self.obstacle_rows
        if np.shape(tables) != np.shape(mine) or not np.array_equal(np.asarray(tables), np.asarray(mine)):
            raise ValueError("checkpoint was taken with other obstacles")
        for name, dst in (("state", self.state), ("time", self.time), ("reward", self.reward), ("term", self.term),
                          ("done_bits", self.done_bits)):
            src = d[name]
            if tuple(src.shape) != tuple(dst.shape) or src.dtype != dst.dtype:
                raise ValueError("checkpoint tensor %s is %s %s, expected %s %s" % (name, tuple(src.shape), src.dtype, tuple(dst.shape), dst.dtype))
            dst.copy_(src)
        if self.obs_norm_buf is not None and "obs_norm" in d:
            self.obs_norm_buf.copy_(d["obs_norm"])
        if self.clipped_actions is not None and "clipped_actions" in d:
            self.clipped_actions.fill_(int(d["clipped_actions"]))
        self._tick, self._resets = int(d["tick"]), int(d["resets"])
        self._device_tick = -1            # graphs captured on this batch read the tick base from the device: refresh it
        return self

    def snapshot(self):
        """The batch as it is now, kept ON THE DEVICE (clones of the state rows, time markers, last outputs and counters):
        restore() puts it back bit for bit.  What state_dict() is for a checkpoint on disk, this is for a look-ahead that
        is taken back (a planner's trial rollouts, bench.py's untimed first replay of a graph)."""
        snap = {name: getattr(self, name).clone() for name in ("state", "time", "reward", "term", "done_bits")}
        if self.obs_norm_buf is not None:
            snap["obs_norm_buf"] = self.obs_norm_buf.clone()
        if self.clipped_actions is not None:
            snap["clipped_actions"] = self.clipped_actions.clone()
        snap["_tick"], snap["_resets"] = self._tick, self._resets
        return snap

    def restore(self, snap):
        """Back to a snapshot() of THIS batch (device-to-device copies on the current stream, no synchronisation)."""
        for name in ("state", "time", "reward", "term", "done_bits", "obs_norm_buf", "clipped_actions"):
            if name in snap:
                getattr(self, name).copy_(snap[name])
        self._tick, self._resets = snap["_tick"], snap["_resets"]
        self._device_tick = -1            # graphs captured on this batch read the tick base from the device: refresh it
        self._sync_device_tick()
        return self

    # ------------------------------------------------------------------ helpers
    def done_mask(self):
        """uint8 [N] done flags unpacked from the ballot words (for checks; step() already returns term)."""
        torch = self.torch
        words = self.done_bits[: (self.num_envs + 63) // 64]
        shifts = torch.arange(64, device=self.device, dtype=torch.int64)
        bits = (words.unsqueeze(1) >> shifts) & 1
        return bits.reshape(-1)[: self.num_envs].to(torch.uint8)

    def set_state(self, state7, time=None, soa=False):
        """Overwrite the worlds' state (teacher forcing in tests): float [N][7], or [7][N] with soa=True."""
        torch = self.torch
        s = torch.as_tensor(np.asarray(state7, dtype=np.float32)) if not isinstance(state7, torch.Tensor) else state7
        s = s.to(device=self.device, dtype=torch.float32)
        if not soa:
            if s.shape != (self.num_envs, 7):
                raise ValueError("state must be [N][7]")
            s = s.t()
        if s.shape != (7, self.num_envs):
            raise ValueError("state must be [7][N]")
        self.state[:, : self.num_envs].copy_(s)
        if time is not None:
            t = torch.as_tensor(np.asarray(time, dtype=np.int32)) if not isinstance(time, torch.Tensor) else time
            self.time[: self.num_envs].copy_(t.to(device=self.device, dtype=torch.int32))
