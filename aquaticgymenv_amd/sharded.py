"""Range partition of the world batch over the GPUs of one node + the one collective of the path.

The worlds are independent (no cross-world term anywhere in gym_aqua/envs/aqua.py:135-213), so the
batch shards with no data-path exchange: rank r owns the global worlds [offset_r, offset_r + count_r)
and passes offset_r as `env_offset`; the Philox streams are keyed by the global index, so the union
of the shards is bit-identical to a single-device run (tests/test_hip_parity.py::test_shard_invariance).

The only exchange is the episodic done mask: every rank contributes its bit-packed ballot words
(uint64 per 64 worlds, 32 KiB per step at 262 144 worlds) and receives everybody's.  One process per
GPU, torch.distributed; backend "nccl" is RCCL over xGMI on MI355X, "gloo" is used by the CPU tests.
The exchange runs on side streams and is consumed late (it is not on the next step's dependency
chain -- restarts are local), so the step stream never waits for it.  Two transports (DoneMaskExchange):
RCCL's all-gather, or peer copies into receive buffers mapped through hipIpcMemHandle -- no kernel on
the compute units, which are exactly one round of step blocks at 262 144 worlds per GPU.
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


class _DevicePointerArray(object):
    """a device allocation the library owns (AquaIpcBuffer), shown to torch through __cuda_array_interface__"""

    def __init__(self, ptr, shape, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def ipc_block_offset(slot, sender, world, steps, words, itemsize=8):
    """byte offset of `sender`'s [steps][words] block inside slot `slot` of a receive buffer [slots][world][steps][words]"""
    if not (0 <= sender < world) or slot < 0:
        raise ValueError("bad slot %d / sender %d of %d" % (slot, sender, world))
    return ((slot * world + sender) * steps * words) * itemsize


class DoneMaskExchange(object):
    """Every rank's [steps][words] int64 done-mask block delivered to every rank of `group`: gathered[slot] is
    [world][steps][words].  words must be the same on every rank (pad the last shard).

    kind="rccl" (default, what BASELINE.json's north_star names): torch.distributed.all_gather_into_tensor -- RCCL over
        xGMI on device tensors, on a private side stream; gloo on CPU tensors, synchronously.
    kind="ipc": no collective kernel.  Every rank owns a receive buffer (include/aqua_hip.h aqua_ipc_*), the ranks exchange
        its 64-byte handle ONCE over the process group and map each other's buffers; a block is published as world - 1
        asynchronous device-to-device copies (one side stream per peer, so the copies of a block use different links)
        plus a local copy.  copy_engine: "waves" = a short kernel of single-wavefront workgroups, "dma" = hipMemcpyAsync
        (the copy engines over xGMI: nothing on the compute units), "auto" (default) = waves into the own buffer, dma
        into the peers'.  There is no per-block handshake: a rank may publish at most `slots` blocks between two
        fence() calls, and what the OTHER ranks sent is complete in gathered[slot] after the next fence() (finish() +
        barrier: bench.py's region boundary).  Setting it up and probing it are collective decisions (_agree()).
    (Round 2 measured 18-21 % of the step stream for the RCCL gather on one rank and blamed RCCL's workgroups; the cost was
    the side stream's device-side wait on the step stream, which every transport shared: with the host-side pump below
    one rank pays 0.3-1.7 % for either kind, profiles/r03/exchange_overhead_one_rank.txt.)
    Either kind: the step stream never waits for the exchange, except before it overwrites a source buffer whose last
    copy has not been read yet (wait_source) -- and no stream of the device ever waits for the step stream: the copies
    are queued by a pump thread once the block's event has completed (see gather_async)."""

    def __init__(self, steps, words, device, group=None, double_buffer=True, kind="rccl", slots=None, copy_engine="auto"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.collective = dist.is_available() and dist.is_initialized()   # a 1-rank group still runs the collective
        self.world = dist.get_world_size(group) if self.collective else 1
        self.rank = dist.get_rank(group) if self.collective else 0
        self.steps, self.words = int(steps), int(words)
        self.device = torch.device(device)
        if kind not in ("rccl", "ipc"):
            raise ValueError("kind must be 'rccl' or 'ipc'")
        self.kind = kind
        if copy_engine not in ("auto", "waves", "dma"):
            raise ValueError("copy_engine must be 'auto', 'waves' or 'dma'")
        self.copy_engine = copy_engine
        nbuf = int(slots) if slots is not None else (2 if double_buffer else 1)
        if nbuf < 1:
            raise ValueError("slots must be >= 1")
        self.slots = nbuf
        self._slot = 0
        self._pending = [None] * nbuf
        self._source_busy = {}            # source_id -> the job that reads that source
        self._since_fence = 0
        self._ipc = None
        self._pump = None
        self._deferred = {}               # slot -> job whose copies have not been handed to the pump yet
        self._side_used = False           # the pump has queued work on the side streams since the last finish()
        if kind == "ipc":
            self._setup_ipc()
        else:
            self.gathered = [torch.zeros((self.world, self.steps, self.words), dtype=torch.int64, device=self.device)
                             for _ in range(nbuf)]
            self._side = [torch.cuda.Stream(device=self.device)] if self.device.type == "cuda" else []

    # ------------------------------------------------------------------ kind="ipc"
    def _agree(self, error, what):
        """Every rank reports whether `what` worked for it; if it failed ANYWHERE, every rank raises -- together, so that no
        rank goes on to a collective the others never reach (a one-sided fallback would hang the job)."""
        if not self.collective:
            if error is not None:
                raise RuntimeError("%s failed: %s: %s" % (what, type(error).__name__, error))
            return
        mine = None if error is None else "%s: %s" % (type(error).__name__, error)
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, mine, group=self.group)
        failed = ["rank %d: %s" % (r, m) for r, m in enumerate(everyone) if m is not None]
        if failed:
            raise RuntimeError("%s failed on %d of %d ranks (%s)" % (what, len(failed), self.world, "; ".join(failed)))

    def _setup_ipc(self):
        import ctypes
        from . import _capi
        torch, dist = self.torch, self.dist
        if self.device.type != "cuda":
            raise RuntimeError("DoneMaskExchange(kind='ipc') moves device buffers between GPU processes: device must be a HIP device")
        lib = _capi.lib
        nbytes = self.slots * self.world * self.steps * self.words * 8
        buf, base, raw, err = ctypes.c_void_p(), 0, b"", None
        try:                                   # (1) the own receive buffer and its handle: local, nothing collective in here
            with torch.cuda.device(self.device):
                _capi.check(lib.aqua_ipc_buffer_create(nbytes, ctypes.byref(buf)), "aqua_ipc_buffer_create")
                handle = ctypes.create_string_buffer(_capi.IPC_HANDLE_BYTES)
                _capi.check(lib.aqua_ipc_buffer_handle(buf, handle), "aqua_ipc_buffer_handle")
                base, raw = int(lib.aqua_ipc_buffer_ptr(buf)), bytes(handle.raw)
        except Exception as exc:
            err = exc
        handles = [raw]
        if self.collective:                    # (2) every rank takes part, whatever happened to it in (1)
            handles = [None] * self.world
            dist.all_gather_object(handles, raw, group=self.group)
        peers = [None] * self.world
        if err is None:
            try:                               # (3) map the others: local again
                peers[self.rank] = base
                with torch.cuda.device(self.device):
                    for r in range(self.world):
                        if r != self.rank:
                            if not handles[r]:
                                raise RuntimeError("rank %d exported no buffer" % r)
                            p = ctypes.c_void_p()
                            _capi.check(lib.aqua_ipc_open(handles[r], ctypes.byref(p)), "aqua_ipc_open (rank %d)" % r)
                            peers[r] = int(p.value)
            except Exception as exc:
                err = exc
        try:
            self._agree(err, "mapping the done-mask receive buffers (hipIpcMemHandle)")     # (4) all or nobody
        except Exception:
            for r, p in enumerate(peers):
                if r != self.rank and p:
                    lib.aqua_ipc_close(p)
            if buf.value:
                lib.aqua_ipc_buffer_destroy(buf)
            raise
        with torch.cuda.device(self.device):
            whole = torch.as_tensor(_DevicePointerArray(base, (self.slots, self.world, self.steps, self.words)), device=self.device)
        import ctypes as _ct
        fanout = [(_ct.c_void_p * self.world)(*[p + ipc_block_offset(k, self.rank, self.world, self.steps, self.words) for p in peers])
                  for k in range(self.slots)]       # per slot: where this rank's block goes in every rank's buffer
        self._ipc = {"lib": lib, "buf": buf, "peers": peers, "whole": whole, "fanout": fanout}
        self.gathered = [whole[s] for s in range(self.slots)]
        # one side stream per destination (the own slot included): the copies of one block run side by side
        self._side = [torch.cuda.Stream(device=self.device) for _ in range(self.world)]

    def probe(self):
        """kind="ipc": one block with a rank-specific pattern through the whole path (pump, side streams, every peer), then
        every rank checks every block.  Raises on ALL ranks if it failed on any (bench.py then falls back to RCCL)."""
        torch = self.torch
        base = torch.arange(self.steps * self.words, dtype=torch.int64, device=self.device).reshape(self.steps, self.words)
        err, slot, slot2 = None, 0, 0
        try:                                   # both publish paths: pump + side streams, then the in-stream fan-out launch
            slot = self.gather_async(base + (self.rank + 1) * 1000003)
            slot2 = self.gather_async(base - (self.rank + 1) * 7919, final=True)
        except Exception as exc:
            err = exc
        try:
            self.finish()
        except Exception as exc:
            err = err or exc
        if self.collective:
            self.dist.barrier(group=self.group)
        self._since_fence = 0
        if err is None:
            try:
                for r in range(self.world):
                    if not torch.equal(self.gathered[slot][r], base + (r + 1) * 1000003) \
                            or not torch.equal(self.gathered[slot2][r], base - (r + 1) * 7919):
                        raise RuntimeError("the probe blocks of rank %d did not arrive intact" % r)
            except Exception as exc:
                err = exc
        self._agree(err, "the probe exchange through the mapped buffers")

    def close(self):
        """unmap the peers' buffers and free the own one (after a fence(): nobody may still be writing into it)"""
        self.finish()
        self.stop()
        if self._ipc is None:
            return
        if self.collective:
            self.dist.barrier(group=self.group)
        lib, ipc = self._ipc["lib"], self._ipc
        self.gathered = []
        ipc["whole"] = None
        for r, p in enumerate(ipc["peers"]):
            if r != self.rank and p:
                lib.aqua_ipc_close(p)
        lib.aqua_ipc_buffer_destroy(ipc["buf"])
        self._ipc = None

    # ------------------------------------------------------------------ both kinds
    def _job_events(self, job):
        """host: until the pump has queued the job's copies (the GPU has reached the block's last step); then its events"""
        if job is None:
            return ()
        for s, j in list(self._deferred.items()):      # somebody needs it now: it cannot stay deferred
            if j is job:
                self.publish(s)
        job.submitted.wait()
        if job.error is not None:
            raise job.error
        return job.events

    def _wait_events(self, events):
        if not events:
            return
        cur = self.torch.cuda.current_stream(self.device)
        for ev in events:
            if not ev.query():            # (a wait for a finished event still costs the stream a barrier packet)
                cur.wait_event(ev)

    def wait_source(self, source_id):
        """Make the current stream wait until the exchange that last READ the caller's buffer `source_id` is done
        (call before the kernels that overwrite that buffer)."""
        self._wait_events(self._job_events(self._source_busy.pop(source_id, None)))

    def gather_async(self, local_bits, source_id=None, final=False, defer=False):
        """Queue the exchange of local_bits ([steps][words] int64, contiguous).  Returns the slot index whose
        `gathered[slot]` holds the result: after wait(slot) for kind="rccl", after the next fence() for kind="ipc".
        final=True: nothing follows this block on the producing stream before the caller drains it (the last block of a
        timed region).  kind="ipc" then delivers it with ONE fan-out launch on the producing stream itself, in stream
        order behind the block's last step: no host round trip, no second stream, no thread hand-off.
        defer=True: only the block's completion event is recorded now; the copies are queued by publish(slot) -- which a
        caller that works in short bursts calls at the START of its next burst, so that the block travels while the next
        steps run (the mask is consumed one block late, as SURVEY.md section 8e plans it) instead of behind an idle GPU."""
        torch, dist = self.torch, self.dist
        if tuple(local_bits.shape) != (self.steps, self.words) or local_bits.dtype != torch.int64 \
                or not local_bits.is_contiguous():
            raise ValueError("local_bits must be a contiguous int64 [%d][%d] tensor" % (self.steps, self.words))
        if self.kind == "ipc" and self._since_fence >= self.slots:
            raise RuntimeError("DoneMaskExchange(kind='ipc'): %d blocks published since the last fence(), the receive "
                               "buffers hold %d" % (self._since_fence, self.slots))
        slot = self._slot
        self._slot = (self._slot + 1) % self.slots
        self._since_fence += 1
        self.wait(slot)                       # the buffer we are about to overwrite must have been consumed
        out = self.gathered[slot]
        if not self._side:                    # CPU tensors (gloo): synchronously
            if self.collective:
                dist.all_gather_into_tensor(out.view(-1), local_bits.view(-1), group=self.group)
            else:
                out[0].copy_(local_bits)
            return slot
        # Device tensors.  The block's copies go to side streams, but NOT behind a device-side wait: a stream that sits on
        # an unfinished event of the step stream costs every step launch of the other stream ~1 us for as long as it waits
        # (profiles/r03/exchange_pieces.txt: 5.02 -> 6.02 us per step with nothing but that wait -- which is what made
        # every transport, RCCL included, look equally expensive).  The wait happens on the HOST instead: a pump thread
        # sleeps in hipEventSynchronize and queues the copies once the block is complete.
        if self.kind == "ipc" and final:
            # one launch on the producing stream.  Everything this rank does with the source buffer or with the slot afterwards
            # is queued behind that launch (later step launches: same stream; later copies: behind a later event of that
            # stream), so the job needs no event of its own -- the host path of a 100-microsecond region stays short
            import ctypes
            from . import _capi
            cur = torch.cuda.current_stream(self.device)
            _capi.check(self._ipc["lib"].aqua_copy_fanout_async(self._ipc["fanout"][slot], self.world, local_bits.data_ptr(),
                                                                self.steps * self.words * 8, ctypes.c_void_p(cur.cuda_stream)),
                        "aqua_copy_fanout_async")
            self._pending[slot] = None
            self._source_busy.pop(source_id, None)
            return slot
        job = _ExchangeJob(torch.cuda.current_stream(self.device).record_event())
        if self.kind == "ipc":
            import ctypes
            from . import _capi
            lib, peers = self._ipc["lib"], self._ipc["peers"]
            nbytes = self.steps * self.words * 8

            def submit():
                for r in range(self.world):
                    st = self._side[r]
                    dst = peers[r] + ipc_block_offset(slot, self.rank, self.world, self.steps, self.words)
                    dma = self.copy_engine == "dma" or (self.copy_engine == "auto" and r != self.rank)
                    _capi.check(lib.aqua_copy_async(dst, local_bits.data_ptr(), nbytes,
                                                    _capi.COPY_ENGINE_DMA if dma else _capi.COPY_ENGINE_WAVES,
                                                    ctypes.c_void_p(st.cuda_stream)), "aqua_copy_async (to rank %d)" % r)
                    job.events.append(st.record_event())
        else:
            side = self._side[0]

            def submit():
                with torch.cuda.stream(side):
                    if self.collective:
                        dist.all_gather_into_tensor(out.view(-1), local_bits.view(-1), group=self.group)
                    else:
                        out[0].copy_(local_bits, non_blocking=True)
                job.events.append(side.record_event())
        job.submit = submit
        job.keep = local_bits                  # the source stays alive until its copies have been queued
        self._pending[slot] = job
        if source_id is not None:
            self._source_busy[source_id] = job
        if defer:
            self._deferred[slot] = job
        else:
            self._pump_put(job)
        return slot

    def publish(self, slot=None):
        """hand the deferred block of `slot` (all deferred blocks when None) to the pump"""
        for s in (list(self._deferred) if slot is None else [slot]):
            job = self._deferred.pop(s, None)
            if job is not None:
                self._pump_put(job)

    def _pump_put(self, job):
        import queue
        import threading
        if self._pump is None:
            self._jobs = queue.Queue()

            def pump():
                self.torch.cuda.set_device(self.device)
                while True:
                    j = self._jobs.get()
                    if j is None:
                        return
                    try:
                        j.ready.synchronize()      # host-side wait (hipEventSynchronize; the GIL is released)
                        self._side_used = True
                        j.submit()
                    except BaseException as exc:   # handed to whoever waits for the job
                        j.error = exc
                    j.submitted.set()
            self._pump = threading.Thread(target=pump, name="done-mask-exchange-pump", daemon=True)
            self._pump.start()
        self._jobs.put(job)

    def wait(self, slot=None):
        """Make the current stream wait for what THIS rank queued for `slot` (all slots when None)."""
        slots = range(self.slots) if slot is None else (slot,)
        for s in slots:
            self._wait_events(self._job_events(self._pending[s]))
            self._pending[s] = None

    def finish(self):
        """Host-side completion of everything this rank queued AND handed to the pump (end of a timed region); a block
        that is still deferred stays deferred."""
        deferred = set(id(j) for j in self._deferred.values())
        for s in range(self.slots):
            if self._pending[s] is not None and id(self._pending[s]) not in deferred:
                self._wait_events(self._job_events(self._pending[s]))
                self._pending[s] = None
        for job in list(self._source_busy.values()):
            if id(job) not in deferred:
                self._job_events(job)
        if self._side_used:
            self._side_used = False
            for st in self._side:
                st.synchronize()

    def fence(self):
        """finish() + a barrier over the group: every block published before it, by ANY rank, is in place in gathered[];
        the slots may be published into again."""
        self.finish()
        if self.collective:
            self.dist.barrier(group=self.group)
        self._since_fence = 0

    def note_fence(self):
        """the caller has run finish() and a barrier of its own (bench.py's region boundary)"""
        self._since_fence = 0

    def stop(self):
        """end the pump thread (after finish())"""
        if self._pump is not None:
            self._jobs.put(None)
            self._pump.join(timeout=10)
            self._pump = None


class _ExchangeJob(object):
    """one published block: `ready` (event of the producing stream), what to queue once it has completed, and the events
    of what was queued"""

    def __init__(self, ready):
        import threading
        self.ready, self.submit, self.keep = ready, None, None
        self.events, self.error = [], None
        self.submitted = threading.Event()


def unpack_done_words(words, count):
    """numpy uint64/int64 words -> uint8 [count] flags (bit i%64 of word i//64)."""
    w = np.ascontiguousarray(words).view(np.uint64).reshape(-1)
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")
    return bits[:count].astype(np.uint8)
