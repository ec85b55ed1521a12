"""Device-resident experience ring for a batch of worlds.

Reference being replaced: the DQN's experience buffer, main/impl/dqn.py:174 (`exp_buffer.append([state,
pred_action, reward, next_state, done])`, a Python deque of per-step lists) and its sampler, dqn.py:251-260
(`random.sample` + `np.vstack`).  Here one batched step appends one transition per world, struct-of-arrays,
straight from the environment's device buffers -- nothing crosses PCIe:

    s     float32 [5][capacity]   normalised observation before the step (main/impl/utils.py:15-33)
    a     uint8   [capacity]      discrete action          | float32 [2][capacity] continuous thrusts
    r     float32 [capacity]
    s2    float32 [5][capacity]   normalised observation after the step
    d     uint8   [capacity]      termination code (0 none, 1 collided, 2 time, 3 success); done = d != 0
    ok    uint8   [capacity]      1 for a real transition, 0 for a world that was restarting (next-step mode: that
                                  step reports reward 0 / term 0 and is not an experience)

The batch lands in consecutive slots (cursor .. cursor + N - 1, modulo capacity), so both sides of the copy are
coalesced; the copies are the C ABI's aqua_ring_write_* kernels on the environment's stream.
"""

from . import _capi


class ReplayRing(object):
    def __init__(self, env, capacity):
        torch = env.torch
        if env.obs_norm_buf is None:
            raise RuntimeError("ReplayRing stores the normalised observation: construct the env with normalized_obs=True")
        if capacity < env.num_envs:
            raise ValueError("capacity (%d) must hold at least one batched step (%d worlds)" % (capacity, env.num_envs))
        self.env = env
        self.capacity = int(capacity)
        dev = env.device
        c = self.capacity
        self.s = torch.zeros((5, c), dtype=torch.float32, device=dev)
        self.s2 = torch.zeros((5, c), dtype=torch.float32, device=dev)
        self.r = torch.zeros(c, dtype=torch.float32, device=dev)
        self.d = torch.zeros(c, dtype=torch.uint8, device=dev)
        self.ok = torch.zeros(c, dtype=torch.uint8, device=dev)
        self.a = torch.zeros((2, c), dtype=torch.float32, device=dev) if env.continuous else \
            torch.zeros(c, dtype=torch.uint8, device=dev)
        self._live = torch.zeros(env.ld, dtype=torch.uint8, device=dev)
        self.cursor = 0          # next slot
        self.size = 0            # filled slots (<= capacity)
        self._open = False

    # ------------------------------------------------------------------ writing
    def _f32(self, ring, src, src_ld, rows):
        e = self.env
        _capi.check(_capi.lib.aqua_ring_write_f32(ring.data_ptr(), self.capacity, self.capacity, self.cursor, src.data_ptr(),
                                                  src_ld, rows, e.num_envs, e._stream()), "aqua_ring_write_f32")

    def _u8(self, ring, src, src_ld, rows):
        e = self.env
        _capi.check(_capi.lib.aqua_ring_write_u8(ring.data_ptr(), self.capacity, self.capacity, self.cursor, src.data_ptr(),
                                                 src_ld, rows, e.num_envs, e._stream()), "aqua_ring_write_u8")

    def before_step(self, action):
        """Record (s, a) of the step about to be taken.  `action`: the tensor that will be passed to env.step()
        (uint8 [N] / float32 [2][ld] soa for continuous worlds)."""
        e = self.env
        torch = e.torch
        with torch.cuda.device(e.device):
            self._f32(self.s, e.obs_norm_buf, e.ld, 5)
            if e.continuous:
                if action.dim() != 2 or action.shape[0] != 2 or action.dtype != torch.float32 or action.stride(1) != 1:
                    raise ValueError("continuous actions are recorded from a float32 [2][>=N] tensor")
                self._f32(self.a, action, action.stride(0), 2)
            else:
                if action.dtype != torch.uint8 or action.dim() != 1 or action.numel() < e.num_envs:
                    raise ValueError("discrete actions are recorded from a uint8 [>=N] tensor")
                self._u8(self.a, action, action.numel(), 1)
            # worlds that are about to be restarted instead of stepped (next-step mode) are not experiences
            self._live.copy_((e.time >= 0) | (e.time <= -3))      # time markers: include/aqua_hip.h
            self._u8(self.ok, self._live, e.ld, 1)
        self._open = True

    def after_step(self, reward=None, term=None):
        """Record (r, s', d) of the step just taken into the same slots and advance the cursor."""
        if not self._open:
            raise RuntimeError("after_step() without before_step()")
        e = self.env
        reward = e.reward if reward is None else reward
        term = e.term if term is None else term
        with e.torch.cuda.device(e.device):
            self._f32(self.r, reward, reward.numel(), 1)
            self._f32(self.s2, e.obs_norm_buf, e.ld, 5)
            self._u8(self.d, term, term.numel(), 1)
        self.cursor = (self.cursor + e.num_envs) % self.capacity
        self.size = min(self.capacity, self.size + e.num_envs)
        self._open = False

    # ------------------------------------------------------------------ reading (dqn.py:251-260)
    def sample(self, batch_size, generator=None):
        """-> (s [B,5], a [B] | [B,2], r [B], s2 [B,5], done bool [B]) of uniformly drawn real transitions."""
        torch = self.env.torch
        if self.size == 0:
            raise RuntimeError("the ring is empty")
        idx = torch.randint(0, self.size, (int(batch_size) * 2,), device=self.env.device, generator=generator)
        idx = idx[self.ok[idx] != 0][:int(batch_size)]          # restarting worlds are < 2 % of the slots
        a = self.a[:, idx].t() if self.env.continuous else self.a[idx]
        return self.s[:, idx].t(), a, self.r[idx], self.s2[:, idx].t(), self.d[idx] != 0
