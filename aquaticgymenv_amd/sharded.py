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
import os
import sys
import threading
import time

import numpy as np


# ------------------------------------------------------------------ bounded time for everything N > 1 sets up
class Watchdog(object):
    """Hard deadline around a stage that may STALL instead of failing (a cross-device hipIpcOpenMemHandle, a copy into a
    peer-mapped buffer, a rendezvous): when `seconds` pass before cancel(), one line naming the stage goes to stderr and
    the process ends with status EXIT_STATUS -- os._exit, never a re-exec (a process that has initialised HIP must not
    replace itself) and never an exception (the stalled call would not see it).  Whoever started the process (bench.py's
    self-launch, a launcher, the driver) gets a non-zero status within a bounded time instead of a hang."""

    EXIT_STATUS = 3

    def __init__(self, seconds, stage, on_expire=None, status=None):
        """status: the exit status announced and handed to on_expire (default EXIT_STATUS; bench.py's optional legs end a
        run whose line is already safe with 0)"""
        self.stage, self.seconds = stage, float(seconds)
        self.status = self.EXIT_STATUS if status is None else int(status)
        self._on_expire = on_expire
        self._timer = threading.Timer(self.seconds, self._expire)
        self._timer.daemon = True
        self._timer.start()

    def _expire(self):
        msg = "aquaticgymenv_amd: deadline of %g s passed in stage '%s' (rank %s): exiting with status %d\n" % (
            self.seconds, self.stage, os.environ.get("RANK", "0"), self.status)
        try:
            sys.stderr.write(msg)
            sys.stderr.flush()
        finally:
            (self._on_expire or os._exit)(self.status)

    def cancel(self):
        self._timer.cancel()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.cancel()
        return False


def injected_faults(environ=None):
    """AQUA_TEST_EXCHANGE_FAIL=<comma list>: fault injection for the tests of the N > 1 set-up path, honoured here only.
         open        mapping a peer's receive buffer fails on the highest rank (as hipIpcOpenMemHandle would)
         probe       the probe block of the highest rank arrives damaged
         stall       mapping a peer's buffer never returns (the soft deadline of open_exchange() abandons it)
         rccl        the RCCL transport fails to set up
         slow        the IPC set-up of the highest rank returns late (AQUA_TEST_SLOW_S seconds, default 6): past a shorter
                     soft deadline its thread must stop at the next stage boundary, not finish beside the next transport
         ab-stall    (bench.py) the set-up of an extra leg behind the main regions never returns on the highest rank
         hard-stall  the caller of open_exchange() itself stalls for kind auto / ipc (only the Watchdog ends that)"""
    raw = (os.environ if environ is None else environ).get("AQUA_TEST_EXCHANGE_FAIL", "")
    return set(w.strip() for w in raw.split(",") if w.strip())


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
        xGMI on device tensors, on a private side stream; gloo on CPU tensors, synchronously.  On device tensors the
        collective is issued by the pump thread (below), at a moment set by GPU progress -- so it runs on a PRIVATE
        process group (dist.new_group(), made here when `group` is None): RCCL wants the same order of collectives on
        every rank of a communicator, and the caller's own collectives (barriers, all-reduces) on the caller's group
        would otherwise interleave with it differently from rank to rank.  A caller that passes its own `group` must
        run no other collective on it between gather_async() and finish().
    kind="ipc": no collective kernel.  Every rank owns a receive buffer (include/aqua_hip.h aqua_ipc_*), the ranks exchange
        its 64-byte handle ONCE over the process group and map each other's buffers; a block is published as world - 1
        asynchronous device-to-device copies (one side stream per peer, so the copies of a block use different links)
        plus a local copy.  copy_engine: "waves" = a short kernel of single-wavefront workgroups, "dma" = hipMemcpyAsync
        (the copy engines over xGMI: nothing on the compute units), "auto" (default) = waves into the own buffer, dma
        into the peers'.  There is no per-block handshake; the contract is by WINDOWS: a window is what lies between two
        fence() calls (finish() + barrier: bench.py's region boundary).  A rank may publish at most `slots` blocks per
        window; the blocks of window w -- everybody's -- are complete in gathered[slot] after the fence that ends w and
        stay readable until the fence that ends window w + 1: the receive buffer holds 2 x slots blocks per sender and
        the windows alternate between its halves, so a fast rank that publishes right behind a fence writes into the half
        nobody is reading.  Setting it up and probing it are collective decisions (_agree()).
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
        self.slots = nbuf                  # kind="ipc": blocks per window; kind="rccl": result buffers in rotation
        self._nslots = 2 * nbuf if kind == "ipc" else nbuf
        self._slot = 0                    # kind="rccl": next buffer in rotation
        self._window, self._in_window = 0, 0      # kind="ipc": the half being published into, blocks published into it
        self._last = None                 # slot of the block published last
        self._pending = [None] * self._nslots
        self._source_busy = {}            # source_id -> the job that reads that source
        self._ipc = None
        self._pump = None
        self._wire_group = group          # the group the transport's own collective runs on (kind="rccl")
        self._side_used = False           # the pump has queued work on the side streams since the last finish()
        self._faults = injected_faults()
        self.stage = getattr(self, "stage", "constructing")      # where a set-up that stalls was last seen (open_exchange())
        self._cancel = getattr(self, "_cancel", None)            # threading.Event of open_exchange()'s sandbox, or None
        if kind == "ipc":
            self._setup_ipc()
        else:
            if "rccl" in self._faults:
                raise RuntimeError("injected fault: the RCCL transport fails to set up")
            self.gathered = [torch.zeros((self.world, self.steps, self.words), dtype=torch.int64, device=self.device)
                             for _ in range(nbuf)]
            self._side = [torch.cuda.Stream(device=self.device)] if self.device.type == "cuda" else []
            if self._side and self.collective and group is None:
                self._wire_group = dist.new_group()       # (collective: every rank constructs the exchange)

    # ------------------------------------------------------------------ kind="ipc"
    def _agree(self, error, what):
        """Every rank reports whether `what` worked for it; if it failed ANYWHERE, every rank raises -- together, so that no
        rank goes on to a collective the others never reach (a one-sided fallback would hang the job)."""
        agree(self.dist if self.collective else None, self.group, self.world, error, what)

    def _enter(self, stage):
        """Stage boundary of the set-up / probe: records where a stall would be seen, and is the CANCELLATION POINT of
        open_exchange()'s sandbox -- once the main thread has given the attempt up (soft deadline passed, the ranks agreed
        to move on) a set-up thread that was slow rather than stuck stops HERE, before its next collective or HIP call,
        instead of mapping buffers and copying on streams beside the transport the run went on with."""
        if self._cancel is not None and self._cancel.is_set():
            raise SetupCancelled("cancelled before stage '%s' (last stage reached: '%s')" % (stage, self.stage))
        self.stage = stage

    def _setup_ipc(self):
        import ctypes
        from . import _capi
        torch, dist = self.torch, self.dist
        if "stall" in self._faults and self.rank == self.world - 1:
            self.stage = "injected fault: a set-up call that never returns"
            time.sleep(1.0e6)                  # (a daemon thread of open_exchange(): abandoned, ends with the process)
        if "slow" in self._faults and self.rank == self.world - 1:
            self.stage = "injected fault: a set-up call that returns late"
            time.sleep(float(os.environ.get("AQUA_TEST_SLOW_S", "6")))
        self._enter("starting the IPC set-up")
        if self.device.type != "cuda":
            raise RuntimeError("DoneMaskExchange(kind='ipc') moves device buffers between GPU processes: device must be a HIP device")
        lib = _capi.lib
        nbytes = self._nslots * self.world * self.steps * self.words * 8
        buf, base, raw, err = ctypes.c_void_p(), 0, b"", None
        peers = [None] * self.world

        def release():                         # what this rank holds so far (a failed agreement, a cancelled set-up)
            for r, p in enumerate(peers):
                if r != self.rank and p:
                    lib.aqua_ipc_close(p)
            if buf.value:
                lib.aqua_ipc_buffer_destroy(buf)

        try:
            self._enter("allocating and exporting the receive buffer")
            try:                               # (1) the own receive buffer and its handle: local, nothing collective in here
                with torch.cuda.device(self.device):
                    _capi.check(lib.aqua_ipc_buffer_create(nbytes, ctypes.byref(buf)), "aqua_ipc_buffer_create")
                    handle = ctypes.create_string_buffer(_capi.IPC_HANDLE_BYTES)
                    _capi.check(lib.aqua_ipc_buffer_handle(buf, handle), "aqua_ipc_buffer_handle")
                    base, raw = int(lib.aqua_ipc_buffer_ptr(buf)), bytes(handle.raw)
            except Exception as exc:
                err = exc
            handles = [raw]
            self._enter("exchanging the buffer handles")
            if self.collective:                # (2) every rank takes part, whatever happened to it in (1)
                handles = [None] * self.world
                dist.all_gather_object(handles, raw, group=self.group)
            self._enter("mapping the peers' receive buffers (hipIpcOpenMemHandle)")
            if err is None:
                try:                           # (3) map the others: local again
                    peers[self.rank] = base
                    if self.rank == self.world - 1 and "open" in self._faults:
                        raise RuntimeError("injected fault: hipIpcOpenMemHandle failed")
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
            self._enter("agreeing on the mapping")
            self._agree(err, "mapping the done-mask receive buffers (hipIpcMemHandle)")     # (4) all or nobody
        except BaseException:
            release()
            raise
        with torch.cuda.device(self.device):
            whole = torch.as_tensor(_DevicePointerArray(base, (self._nslots, self.world, self.steps, self.words)), device=self.device)
        import ctypes as _ct
        fanout = [(_ct.c_void_p * self.world)(*[p + ipc_block_offset(k, self.rank, self.world, self.steps, self.words) for p in peers])
                  for k in range(self._nslots)]     # per slot: where this rank's block goes in every rank's buffer
        self._ipc = {"lib": lib, "buf": buf, "peers": peers, "whole": whole, "fanout": fanout}
        self.gathered = [whole[s] for s in range(self._nslots)]
        # one side stream per destination (the own slot included): the copies of one block run side by side
        self._side = [torch.cuda.Stream(device=self.device) for _ in range(self.world)]
        self.stage = "mapped"

    def probe(self):
        """One block with a rank-specific pattern through the whole path -- pump, side streams, every peer (kind="ipc": also
        the in-stream fan-out launch) -- then every rank checks every rank's block.  Raises on ALL ranks if it failed on
        any (bench.py then falls back to the next transport).  Ends with a fence()."""
        torch = self.torch
        base = torch.arange(self.steps * self.words, dtype=torch.int64, device=self.device).reshape(self.steps, self.words)
        damage = 1 if ("probe" in self._faults and self.rank == self.world - 1) else 0
        err, slot, slot2 = None, 0, None
        self._enter("publishing the probe blocks")
        try:                                   # both publish paths: pump + side streams, then the in-stream fan-out launch
            slot = self.gather_async(base + (self.rank + 1) * 1000003 + damage)
            if self.kind == "ipc" and self.slots >= 2:
                slot2 = self.gather_async(base - (self.rank + 1) * 7919, final=True)
        except Exception as exc:
            err = exc
        self._enter("waiting for the probe blocks")
        try:                                   # (final=True leaves the producing stream to the caller: drain it, then fence)
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()
            self.fence()
        except Exception as exc:
            err = err or exc
        self._enter("checking the probe blocks")
        if err is None:
            try:
                for r in range(self.world):
                    if not torch.equal(self.gathered[slot][r], base + (r + 1) * 1000003) \
                            or (slot2 is not None and not torch.equal(self.gathered[slot2][r], base - (r + 1) * 7919)):
                        raise RuntimeError("the probe blocks of rank %d did not arrive intact" % r)
            except Exception as exc:
                err = exc
        self._agree(err, "the probe exchange (%s)" % self.kind)
        self.stage = "probed"

    def close(self):
        """unmap the peers' buffers and free the own one.  Collective (it runs a fence(): nobody may still be writing into a
        buffer that is about to go away)."""
        self.fence()
        self.stop()
        if self._ipc is None:
            return
        lib, ipc = self._ipc["lib"], self._ipc
        self.gathered = []
        ipc["whole"] = None
        for r, p in enumerate(ipc["peers"]):
            if r != self.rank and p:
                lib.aqua_ipc_close(p)
        lib.aqua_ipc_buffer_destroy(ipc["buf"])
        self._ipc = None

    def abandon(self):
        """Give the exchange up WITHOUT a collective and without a HIP call that could queue behind a stalled one (another
        rank failed or stalled while this one set up fine): the pump thread ends, the buffers are left to the process's
        end.  Nothing has been, or will be, published through an abandoned exchange."""
        self.stop()
        self.gathered = []

    # ------------------------------------------------------------------ both kinds
    def _job_events(self, job):
        """host: until the pump has queued the job's copies (the GPU has reached the block's last step); then its events"""
        if job is None:
            return ()
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

    def last_slot(self):
        """slot of the block this rank published last (None before the first)"""
        return self._last

    def _next_slot(self):
        if self.kind == "ipc":
            if self._in_window >= self.slots:
                raise RuntimeError("DoneMaskExchange(kind='ipc'): %d blocks published since the last fence(), a window holds "
                                   "%d" % (self._in_window, self.slots))
            slot = self._window * self.slots + self._in_window
            self._in_window += 1
        else:
            slot = self._slot
            self._slot = (self._slot + 1) % self.slots
        self._last = slot
        return slot

    def gather_async(self, local_bits, source_id=None, final=False):
        """Queue the exchange of local_bits ([steps][words] int64, contiguous).  Returns the slot index whose
        `gathered[slot]` holds the result: after wait(slot) for kind="rccl"; for kind="ipc" from the next fence() until
        the one after it.
        final=True: nothing follows this block on the producing stream before the caller drains it (the last block of a
        timed region).  kind="ipc" then delivers it with ONE fan-out launch on the producing stream itself, in stream
        order behind the block's last step: no host round trip, no second stream, no thread hand-off.
        (Round 3 also had defer=True / publish(): the block held back and sent during the caller's NEXT burst.  It reached
        the fan-out launch's steady state with erratic first regions -- profiles/r03/exchange_pipelined_last_block.txt --
        was never what bench.py did, and its slot accounting across a fence was wrong; removed.)"""
        torch, dist = self.torch, self.dist
        if tuple(local_bits.shape) != (self.steps, self.words) or local_bits.dtype != torch.int64 \
                or not local_bits.is_contiguous():
            raise ValueError("local_bits must be a contiguous int64 [%d][%d] tensor" % (self.steps, self.words))
        slot = self._next_slot()
        self.wait(slot)                       # what this rank last queued for that slot must have run
        out = self.gathered[slot]
        if not self._side:                    # CPU tensors (gloo): synchronously, on the calling thread
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
                    if self.collective:       # (on the private group: see the class docstring)
                        dist.all_gather_into_tensor(out.view(-1), local_bits.view(-1), group=self._wire_group)
                    else:
                        out[0].copy_(local_bits, non_blocking=True)
                job.events.append(side.record_event())
        job.submit = submit
        job.keep = local_bits                  # the source stays alive until its copies have been queued
        self._pending[slot] = job
        if source_id is not None:
            self._source_busy[source_id] = job
        self._pump_put(job)
        return slot

    def _pump_put(self, job):
        import queue
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
        slots = range(self._nslots) if slot is None else (slot,)
        for s in slots:
            self._wait_events(self._job_events(self._pending[s]))
            self._pending[s] = None

    def finish(self):
        """Host-side completion of everything this rank queued (end of a timed region)."""
        for s in range(self._nslots):
            if self._pending[s] is not None:
                self._wait_events(self._job_events(self._pending[s]))
                self._pending[s] = None
        for job in list(self._source_busy.values()):
            self._job_events(job)
        if self._side_used:
            self._side_used = False
            for st in self._side:
                st.synchronize()

    def fence(self):
        """finish() + a barrier over the group: every block published in the window it ends, by ANY rank, is in place in
        gathered[] and stays there until the NEXT fence; publishing goes on in the other half of the receive buffers."""
        self.finish()
        if self.collective:
            self.dist.barrier(group=self.group)
        self.note_fence()

    def note_fence(self):
        """the caller has run finish() and a barrier of its own (bench.py's region boundary)"""
        self._window ^= 1
        self._in_window = 0

    def stop(self):
        """end the pump thread (after finish())"""
        if self._pump is not None:
            self._jobs.put(None)
            self._pump.join(timeout=10)
            self._pump = None


def agree(dist, group, world, error, what):
    """Every rank of `group` reports whether `what` worked for it (error: None or an exception); if it failed ANYWHERE,
    every rank raises, together.  dist None: a lone process."""
    if dist is None:
        if error is not None:
            raise RuntimeError("%s failed: %s: %s" % (what, type(error).__name__, error))
        return
    mine = None if error is None else "%s: %s" % (type(error).__name__, error)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    failed = ["rank %d: %s" % (r, m) for r, m in enumerate(everyone) if m is not None]
    if failed:
        raise RuntimeError("%s failed on %d of %d ranks (%s)" % (what, len(failed), world, "; ".join(failed)))


# set-up threads that open_exchange() left behind in a call that had not returned (a process that has any should end with
# os._exit once its work is done: tearing the process groups down under such a thread may not return either)
ABANDONED_SETUP_THREADS = []
# what set-up threads that came back late (after their attempt had been given up) had reached when they stopped
LATE_SETUP_NOTES = []


def open_exchange(kind, steps, words, device, slots=2, copy_engine="auto", soft_deadline_s=45.0, allow_rccl=True):
    """The done-mask exchange of a run, set up so that no failure and no STALL of a transport can take the run with it.
    -> (exchange or None, kind in use: "ipc" | "rccl" | None, note: why an earlier choice was dropped, or None).

      kind "auto": peer copies through IPC-mapped receive buffers if every rank can map every other rank's buffer and a
                   probe block arrives intact everywhere; else RCCL's all-gather (allow_rccl: the process group's backend
                   can move device tensors), probed the same way; else no exchange at all (the note says why).
      kind "ipc" / "rccl": that transport or an exception (raised on every rank together).   kind "none": no exchange.

    The IPC set-up runs in a SANDBOX: a daemon thread, with a process group of its own (gloo) for the handles and the
    agreements.  A rank whose hipIpcOpenMemHandle or first peer copy never returns leaves its thread behind after
    soft_deadline_s; the main threads then agree over the caller's (untouched) default group that the attempt is off, the
    ranks that had finished abandon() what they built, and the run goes on with the next transport -- under the launcher
    the driver uses there is nobody to start fresh children, so the fallback has to happen inside the process.  What a
    soft deadline cannot end (a stalled collective of the default group itself) is the caller's Watchdog's."""
    import torch
    import torch.distributed as dist
    if kind not in ("auto", "ipc", "rccl", "none"):
        raise ValueError("kind must be 'auto', 'ipc', 'rccl' or 'none'")
    if kind == "none":
        return None, None, None
    collective = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if collective else 1
    notes = []
    if kind in ("auto", "ipc"):
        if "hard-stall" in injected_faults():
            time.sleep(1.0e6)
        # (its collectives time out by themselves a little after the soft deadline: a set-up thread that was left behind in
        # one ends instead of sitting there when the process group is torn down)
        from datetime import timedelta
        sandbox_group = dist.new_group(backend="gloo", timeout=timedelta(seconds=soft_deadline_s + 20.0)) if collective else None
        # The main thread and the set-up thread share `box` under `lock`.  `cancel` is set by the main thread once the ranks
        # have agreed that the attempt is off: a thread that was slow rather than stuck then stops at its next stage
        # boundary (DoneMaskExchange._enter), releases what it holds and leaves a note; a thread that finishes the whole
        # set-up AFTER the deadline finds `cancel` set under the lock and abandons its exchange itself -- nothing of an
        # abandoned attempt maps buffers, copies or runs collectives beside the transport the run went on with.
        box, lock, cancel = {}, threading.Lock(), threading.Event()

        def attempt():
            ex = None
            try:
                if torch.device(device).type == "cuda":
                    torch.cuda.set_device(torch.device(device))
                ex = DoneMaskExchange.__new__(DoneMaskExchange)
                ex.stage, ex._cancel = "starting", cancel
                with lock:
                    box["partial"] = ex
                ex.__init__(steps, words, device, group=sandbox_group, kind="ipc", slots=max(2, slots), copy_engine=copy_engine)
                ex.probe()
                with lock:
                    if cancel.is_set():            # finished, but too late: the run has gone on without it
                        raise SetupCancelled("cancelled after stage '%s'" % ex.stage)
                    box["ex"] = ex
            except BaseException as exc:
                if isinstance(exc, SetupCancelled):
                    if ex is not None:
                        try:
                            ex.abandon()
                        except Exception:
                            pass
                    LATE_SETUP_NOTES.append("rank %s: %s" % (os.environ.get("RANK", "0"), exc))
                with lock:
                    box["error"] = exc

        t = threading.Thread(target=attempt, name="done-mask-exchange-setup", daemon=True)
        t.start()
        t.join(soft_deadline_s)
        with lock:                                 # ONE look at the thread's outcome, after the join
            alive = t.is_alive()
            ex_ready, err_seen, partial = box.get("ex"), box.get("error"), box.get("partial")
        if ex_ready is not None:
            mine = None
        elif alive:
            mine = TimeoutError("still in stage '%s' after %g s" % (getattr(partial, "stage", "starting"), soft_deadline_s))
        else:
            mine = err_seen or RuntimeError("the set-up thread ended without a result")
        try:
            agree(dist if collective else None, None, world, mine, "the IPC done-mask exchange")
            return ex_ready, "ipc", None
        except RuntimeError as exc:
            with lock:
                cancel.set()                       # from here on a late thread stops at its next stage boundary
                late = box.get("ex")               # (it may have finished between the look above and now)
            for built in set(e for e in (ex_ready, late) if e is not None):
                built.abandon()
            if t.is_alive():
                ABANDONED_SETUP_THREADS.append(t)
            if kind == "ipc":
                raise
            notes.append("ipc unavailable (%s)" % exc)
    if allow_rccl:
        err, ex = None, None
        try:
            ex = DoneMaskExchange(steps, words, device, kind="rccl")
        except Exception as exc:
            err = exc
        try:
            agree(dist if collective else None, None, world, err, "setting up the RCCL done-mask exchange")
            ex.probe()
            return ex, "rccl", ("; ".join(notes) + ": RCCL all-gather") if notes else None
        except RuntimeError as exc:
            if ex is not None:
                ex.abandon()
            if kind == "rccl":
                raise
            notes.append("rccl unavailable (%s)" % exc)
    elif kind == "rccl":
        raise RuntimeError("the RCCL transport needs a process group whose backend moves device tensors (nccl)")
    return None, None, "; ".join(notes) + ": NO done-mask exchange in this run"


class SetupCancelled(RuntimeError):
    """raised inside a set-up thread of open_exchange() that reaches a stage boundary after the attempt was given up"""


class _ExchangeJob(object):
    """one published block: `ready` (event of the producing stream), what to queue once it has completed, and the events
    of what was queued"""

    def __init__(self, ready):
        self.ready, self.submit, self.keep = ready, None, None
        self.events, self.error = [], None
        self.submitted = threading.Event()


def unpack_done_words(words, count):
    """numpy uint64/int64 words -> uint8 [count] flags (bit i%64 of word i//64)."""
    w = np.ascontiguousarray(words).view(np.uint64).reshape(-1)
    bits = np.unpackbits(w.view(np.uint8), bitorder="little")
    return bits[:count].astype(np.uint8)
