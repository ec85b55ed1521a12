"""Range partition of the world batch over the GPUs of one node + the one collective of the path.

The worlds are independent (no cross-world term anywhere in gym_aqua/envs/aqua.py:135-213), so the
batch shards with no data-path exchange: rank r owns the global worlds [offset_r, offset_r + count_r)
and passes offset_r as `env_offset`; the Philox streams are keyed by the global index, so the union
of the shards is bit-identical to a single-device run (tests/test_hip_parity.py::test_shard_invariance).

The only exchange is the episodic done mask: every rank contributes its bit-packed ballot words
(uint64 per 64 worlds, 32 KiB per step at 262 144 worlds) and receives everybody's.  One process per
GPU, torch.distributed; backend "nccl" is RCCL over xGMI on MI355X, "gloo" is used by the CPU tests.
The gather runs on a side stream and is consumed late (it is not on the next step's dependency
chain -- restarts are local), so the step stream never waits for it.
"""
import numpy as np


def shard_range(total, world_size, rank):
    """Contiguous range partition in units of 64 worlds (one ballot word), remainder to the low ranks.
    Returns (offset, count)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    words = (total + 63) // 64
    base, extra = divmod(words, world_size)
    my_words = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    offset = first * 64
    count = max(0, min(total, offset + my_words * 64) - offset)
    return offset, count


class DoneMaskExchange(object):
    """All-gather of [steps][words] int64 done-mask blocks across the ranks of `group`.

    words must be the same on every rank (pad the last shard); gathered shape is
    [world][steps][words].  On CUDA/HIP tensors the collective is queued on a private side stream
    behind an event of the producing stream; on CPU tensors (gloo) it runs synchronously.
    """

    def __init__(self, steps, words, device, group=None, double_buffer=True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.collective = dist.is_available() and dist.is_initialized()   # a 1-rank group still runs the collective
        self.world = dist.get_world_size(group) if self.collective else 1
        self.rank = dist.get_rank(group) if self.collective else 0
        self.steps, self.words = int(steps), int(words)
        self.device = torch.device(device)
        nbuf = 2 if double_buffer else 1
        self.gathered = [torch.zeros((self.world, self.steps, self.words), dtype=torch.int64, device=self.device)
                         for _ in range(nbuf)]
        self._slot = 0
        self._pending = [None] * nbuf
        self._source_busy = {}            # source_id -> event after which the caller may overwrite that source
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def wait_source(self, source_id):
        """Make the current stream wait until the gather that last READ the caller's buffer `source_id` is done
        (call before the kernels that overwrite that buffer)."""
        ev = self._source_busy.pop(source_id, None)
        if ev is not None:
            self.torch.cuda.current_stream(self.device).wait_event(ev)

    def gather_async(self, local_bits, source_id=None):
        """Queue the all-gather of local_bits ([steps][words] int64, contiguous).  Returns the slot index
        whose `gathered[slot]` holds the result after wait(slot)."""
        torch, dist = self.torch, self.dist
        if tuple(local_bits.shape) != (self.steps, self.words) or local_bits.dtype != torch.int64 \
                or not local_bits.is_contiguous():
            raise ValueError("local_bits must be a contiguous int64 [%d][%d] tensor" % (self.steps, self.words))
        slot = self._slot
        self._slot = (self._slot + 1) % len(self.gathered)
        self.wait(slot)                       # the buffer we are about to overwrite must have been consumed
        out = self.gathered[slot]
        if not self.collective:
            if self._side is not None:
                self._side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._side):
                    out[0].copy_(local_bits, non_blocking=True)
                    local_bits.record_stream(self._side)
                self._pending[slot] = self._side.record_event()
                if source_id is not None:
                    self._source_busy[source_id] = self._pending[slot]
            else:
                out[0].copy_(local_bits)
            return slot
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                dist.all_gather_into_tensor(out.view(-1), local_bits.view(-1), group=self.group)
                local_bits.record_stream(self._side)
            self._pending[slot] = self._side.record_event()
            if source_id is not None:
                self._source_busy[source_id] = self._pending[slot]
        else:
            dist.all_gather_into_tensor(out.view(-1), local_bits.view(-1), group=self.group)
        return slot

    def wait(self, slot=None):
        """Make the current stream wait for the gather in `slot` (all slots when None)."""
        slots = range(len(self.gathered)) if slot is None else (slot,)
        for s in slots:
            ev = self._pending[s]
            if ev is not None:
                self.torch.cuda.current_stream(self.device).wait_event(ev)
                self._pending[s] = None

    def finish(self):
        """Host-side completion of everything queued (end of a timed region)."""
        self.wait()
        if self._side is not None:
            self._side.synchronize()


def unpack_done_words(words, count):
    """numpy uint64/int64 words -> uint8 [count] flags (bit i%64 of word i//64)."""
    w = np.ascontiguousarray(words).view(np.uint64).reshape(-1)
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")
    return bits[:count].astype(np.uint8)
