#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched AquaEnv step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
`python bench.py --gpus N` with N > 1 and no launcher around it starts the second form itself, as a fresh child process,
before this process has made any HIP call, relays rank 0's line and exits with the children's status.

A "step" is ONE batched step (one launch of the fused step kernel) over one batch of worlds.
Workload at N=1 (BASELINE.json configs[2], the configuration the metric is quoted on):
262 144 worlds, discrete uint8 actions (pre-generated, i.i.d. uniform), 4 circle + 4 rectangle
obstacles, waves on, Philox noise, auto-reset on.  N > 1: the same batch PER GPU (weak scaling),
range-partitioned global world indices, the done mask exchanged on side streams (peer copies through hipIpcMemHandle
when the ranks can map each other's buffers, else RCCL's all-gather).

Protocol: W untimed warm-up steps, then timed regions (max(5, ceil(2000 / K)) of them) of EXACTLY K
steps each, every region bracketed by barrier + torch.cuda.synchronize() on both sides (a rank's clock
stops when ITS queue has drained, steps and done-mask gathers; the barrier follows) and reduced
with MAX over the ranks; the MEDIAN region is the one reported (`value`, `ms_per_step`; all of them are
listed under `regions_ms`, the first five are also reported on their own: `roofline.first_regions`).  Steps are queued as replays of captured HIP graphs of min(CHUNK, K) steps
(+ one graph for the remainder), so that any K >= 1 and W >= 0 runs the same way.  Inputs are resident
in HBM before the timed regions.  `roofline` comes from HIP events recorded on the launch stream around each region's
launches.  Rank 0 prints ONE JSON line.

The CPU baselines (`cpu_baseline*`, oracle/ = test infrastructure, never the product path) run BEFORE
the process touches the GPU: their worker processes are started with fork+exec, which must not happen
from a process that has initialised HIP.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

A_DISCRETE = 62      # algorithmic bytes per world-step, SURVEY.md 8(d): 33 read + 29 written
A_CONTINUOUS = 69
HBM_PEAK_GBPS = 8000.0
HBM_COPY_GBPS = 6290.0   # measured float4 copy (MI355X_MICROARCH.md chip table): SURVEY.md 8(d) asks for this fraction too
CHUNK = 100          # steps per captured HIP graph (at most)
GATHER_EVERY = 5     # chunks per done-mask block: one [GATHER_EVERY * CHUNK][words] all-gather per block (N > 1)
REGIONS = 5          # timed regions at least; the median is reported
MIN_TIMED_STEPS = 2000   # ... and at least this many timed steps in all (one default region's worth): see regions_for()
MAX_REGIONS = 400
ROLLBACK_BELOW_REGIONS = 20   # the untimed, rolled-back first replay of the timed graphs is made only for fewer regions than this
AB_MIN_REGIONS = 20      # N > 1: regions of every extra leg (the other transport, the step path alone)
OUTLIER_FACTOR = 1.5     # regions_outliers: regions slower than this x the median


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=262144, help="worlds per GPU")
    ap.add_argument("--continuous", action="store_true", help="configs[3]: continuous actions")
    ap.add_argument("--no-obstacles", action="store_true", help="configs[1]-style: no obstacles")
    ap.add_argument("--no-auto-reset", action="store_true", help="diagnostic only: finished worlds are not restarted")
    ap.add_argument("--reset-mode", type=int, default=2, choices=(1, 2),
                    help="restart of finished worlds: 2 = during the next step (default, fastest), 1 = inside the same launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of EACH CPU baseline leg")
    ap.add_argument("--regions", type=int, default=None,
                    help="timed regions of --steps steps each (median reported); default: max(%d, ceil(%d / steps)), see regions_for()"
                         % (REGIONS, MIN_TIMED_STEPS))
    ap.add_argument("--eager", action="store_true", help="no HIP graph: one C-side launch loop per chunk")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the done-mask exchange (side streams) even on one GPU: rehearsal of the N > 1 path")
    ap.add_argument("--exchange", choices=("auto", "ipc", "rccl", "none"), default="auto",
                    help="transport of the done-mask exchange: peer copies into IPC-mapped receive buffers (no kernel on the "
                         "compute units), RCCL's all-gather, auto = ipc when the ranks can map each other and a probe block "
                         "arrives, else rccl, else none; none = the step path alone (the line says so: "
                         "config.done_mask_exchange false)")
    ap.add_argument("--exchange-note", default=None,
                    help="(set by bench.py's own self-launch when it starts fresh ranks after a failed attempt: what failed; "
                         "copied into config.done_mask_exchange_note)")
    ap.add_argument("--soft-deadline", type=float, default=45.0,
                    help="seconds the IPC set-up + probe may take before every rank drops it and goes on with the next "
                         "transport (sharded.open_exchange)")
    ap.add_argument("--setup-deadline", type=float, default=150.0,
                    help="N > 1: seconds from the rendezvous to the end of the exchange's set-up; past it the rank writes one "
                         "line to stderr and exits with status 3 (sharded.Watchdog) -- a stalled collective or copy becomes a "
                         "bounded failure instead of a hang")
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0, help="N > 1: timeout of init_process_group")
    ap.add_argument("--launch-deadline", type=float, default=330.0,
                    help="self-launch (bare `bench.py --gpus N`): seconds the first set of ranks may take (a fresh box spends "
                         "1-2 minutes importing torch); the fresh ranks of a fallback get 150 s")
    ap.add_argument("--copy-engine", choices=("auto", "waves", "dma"), default="auto",
                    help="--exchange ipc: copy by single-wavefront workgroups, by hipMemcpyAsync, or auto = wavefronts into "
                         "the own buffer and hipMemcpyAsync (the copy engines) into the other GPUs' buffers")
    ap.add_argument("--ranks-on-one-gpu", action="store_true",
                    help="rehearsal on a 1-GPU box: all --gpus ranks share cuda:0, gloo carries the barriers (RCCL refuses two "
                         "ranks on one device), the done mask moves by IPC peer copies.  Checks the N > 1 path on device "
                         "tensors; its timings mean nothing")
    ap.add_argument("--graph-node-events", action="store_true",
                    help="diagnostic: a region that is one graph ALSO carries event-record nodes at its head and tail "
                         "(round 2's clock), reported beside the stream events as roofline.launch_us_graph_nodes_regions")
    ap.add_argument("--region-clock", choices=("stream", "launch"), default="stream",
                    help="the roofline's clock.  stream (default): HIP events recorded on the launch stream around the region's "
                         "graph launches -- two marker packets, 12-14 us per region beyond first kernel start -> last kernel end "
                         "(profiles/r03/burst_timeline.txt).  launch (diagnostic, regions of one block of <= %d steps): the region "
                         "is queued by one C call of plain launches, its events ATTACHED to the first launch (start time) and the "
                         "last one (end time) by hipExtLaunchKernel; the stream events are recorded as well and reported beside "
                         "them.  Measured on one box at --steps 20 (profiles/r03/region_clock.txt): 5.1-5.4 us per step by the "
                         "attached events against 5.5 by the stream events around a graph -- and 13-17 %% less throughput by wall "
                         "clock, noisier regions (the host queues every launch): not the default" % CHUNK)
    ap.add_argument("--first-replay", choices=("upload", "rollback"), default="rollback",
                    help="graphs of the timed region that the warm-up budget cannot replay (warmup < steps): 'upload' = "
                         "hipGraphUpload ahead of region 0; 'rollback' = ALSO one untimed replay whose effect on the batch is "
                         "undone (BatchedAqua.snapshot / restore), so that region 0 is not a graph's first launch")
    ap.add_argument("--ramp-replays", type=int, default=1,
                    help="--first-replay rollback: how many times the untimed replay is repeated before it is taken back")
    ap.add_argument("--settle-us", type=float, default=0.0,
                    help="diagnostic: host pause between the barrier that closes a region and the start of the next region's "
                         "clock (outside every timed interval).  A 20-step graph launched 200 us after a device "
                         "synchronisation runs 0.15-0.3 us per step faster than one launched right behind it, at the price of "
                         "3 %% of wall clock (profiles/r03/k20_settle.txt): off by default")
    ap.add_argument("--no-exchange-ab", action="store_true",
                    help="N > 1 (or --force-exchange): skip the extra legs behind the main regions (extras.exchange_ab: the "
                         "step path alone, and the transport the main regions did NOT use)")
    ap.add_argument("--ab-regions", type=int, default=None,
                    help="regions of each extra leg; default: max(%d, the main regions' rule)" % AB_MIN_REGIONS)
    ap.add_argument("--ab-deadline", type=float, default=120.0,
                    help="seconds all extra legs together may take: past it rank 0 prints the line with what it has and "
                         "extras.exchange_ab.error naming the stage, and every rank exits with status 0 (the main line is "
                         "never lost to an optional leg)")
    ap.add_argument("--teardown-deadline", type=float, default=60.0,
                    help="N > 1: seconds the collective tear-down behind the printed line may take before every rank still in "
                         "it exits with status 0")
    ap.add_argument("--extras", action="store_true", help="also time the fused rollout kernel (separate line)")
    ap.add_argument("--per-world-tables", action="store_true",
                    help="separate line (extras.per_world_tables): every world has its own 8-obstacle list")
    args = ap.parse_args(argv)
    if args.steps < 1:
        ap.error("--steps must be >= 1")
    if args.warmup < 0:
        ap.error("--warmup must be >= 0")
    if args.regions is None:
        args.regions, args.regions_rule = regions_for(args.steps), "max(%d, ceil(%d / steps))" % (REGIONS, MIN_TIMED_STEPS)
    else:
        args.regions_rule = "--regions"
    if args.regions < 1:
        ap.error("--regions must be >= 1")
    if args.ab_regions is None:
        args.ab_regions = max(AB_MIN_REGIONS, regions_for(args.steps))
    if args.ab_regions < 1:
        ap.error("--ab-regions must be >= 1")
    return args


def regions_for(steps):
    """How many timed regions of `steps` steps: five at least, and as many as it takes to time MIN_TIMED_STEPS steps in all.
    Five regions of 20 steps are 0.6 ms of GPU work on a GPU that has been busy for less than a millisecond since the process
    started: they read 0.3-0.5 us per step above what the same command reads a few milliseconds later (60 regions: the first 14
    at 5.1-5.5 us, the rest at 4.7-5.1: profiles/FLOOR.md) -- a sample too short for the median to mean anything.  Every
    region is still exactly `steps` steps between its own brackets and is listed; the first five are also reported on their
    own (roofline.first_regions)."""
    return min(MAX_REGIONS, max(REGIONS, -(-MIN_TIMED_STEPS // max(steps, 1))))


# ------------------------------------------------------------------ the step queue (host logic; tests/test_bench_logic.py)
def plan_region(n_steps, chunk=CHUNK, block_rows=CHUNK * GATHER_EVERY):
    """Segments covering exactly n_steps, starting at done-mask buffer 0, row 0:
    [(buf, row0, steps, gather_after)].  A segment is one graph launch (or one C-side launch loop) of
    `steps` <= chunk steps that writes its done masks to rows [row0, row0 + steps) of buffer `buf`;
    gather_after marks the segment that completes a block (or the region): the block is exchanged behind it."""
    if n_steps < 0 or chunk < 1 or block_rows < chunk:
        raise ValueError("bad plan: n_steps=%d chunk=%d block_rows=%d" % (n_steps, chunk, block_rows))
    segs, buf, row, left = [], 0, 0, n_steps
    while left > 0:
        s = min(chunk, left, block_rows - row)
        row += s
        left -= s
        full = row == block_rows
        segs.append((buf, row - s, s, full or left == 0))
        if full:
            buf, row = buf ^ 1, 0
    return segs


class StepRunner(object):
    """Queues regions of batched steps on `env` (anything with BatchedAqua's rollout()/capture_rollout()).

    Every region starts at buffer 0 / row 0 of the double-buffered done-mask history `hist`, so the segments
    of a region depend on its length only and their graphs can be captured ahead of the timed code
    (prepare()).  With an `exchange` (DoneMaskExchange) each completed block is all-gathered on the side
    stream; the step stream only waits for the gather that last read the buffer it is about to overwrite."""

    def __init__(self, env, actions, hist, exchange=None, use_graph=True, chunk=CHUNK, launch_events=None):
        self.env, self.actions, self.hist, self.exchange = env, actions, hist, exchange
        self.use_graph, self.chunk = use_graph, chunk
        self.launch_events = launch_events               # a LaunchEvents: one-block regions of run(clock=True) carry it
        self.block_rows = int(hist[0].shape[0])
        self.graphs = {}
        self.stream_handle = None                        # set_stream(): the launch stream, marshalled once for graph.launch()
        self._plans = {}
        self.timed = set()
        self.timing_error = None
        self.steps_run = 0

    def set_stream(self, torch_stream):
        """the stream every launch of run() goes to (it must be torch's current stream while run() is called)"""
        import ctypes
        self.stream_handle = ctypes.c_void_p(torch_stream.cuda_stream)

    def plan(self, n_steps):
        segs = self._plans.get(n_steps)                  # (a 100-microsecond region should not pay for re-planning itself)
        if segs is None:
            segs = self._plans[n_steps] = plan_region(n_steps, self.chunk, self.block_rows)
        return segs

    def prepare(self, n_steps, timing=False):
        """capture the graphs a region of n_steps needs; timing=True: a region that is ONE graph carries event-record
        nodes at its head and tail (region_graph_ms())"""
        if not self.use_graph:
            return
        plan = self.plan(n_steps)
        for buf, row0, s, _ in plan:
            key = (buf, row0, s)
            timed = timing and len(plan) == 1
            if key not in self.graphs or (timed and key not in self.timed):
                done = self.hist[buf][row0:row0 + s]
                if timed:
                    try:
                        self.graphs[key] = self.env.capture_rollout(s, actions=self.actions, keep_all=False,
                                                                    done_history=done, timing=True)
                        self.timed.add(key)
                        continue
                    except Exception as exc:         # event-record nodes unsupported: the stream events remain
                        self.timing_error = "%s: %s" % (type(exc).__name__, exc)
                if key not in self.graphs:
                    self.graphs[key] = self.env.capture_rollout(s, actions=self.actions, keep_all=False, done_history=done)

    def upload_unplayed(self, timed_steps, played):
        """hipGraphUpload for the graphs of a region of timed_steps steps that none of the `played` run lengths replays;
        -> how many were uploaded"""
        if not self.use_graph:
            return 0
        seen = set(seg[:3] for n in played for seg in self.plan(n))
        todo = []
        for seg in self.plan(timed_steps):
            if seg[:3] not in seen:
                seen.add(seg[:3])
                todo.append(seg[:3])
        for key in todo:
            self.graphs[key].upload()
        return len(todo)

    def replay_unplayed_and_roll_back(self, timed_steps, played, repeats=1):
        """Replay, once and untimed, the graphs of a region of timed_steps steps that none of the `played` run lengths
        replays, then put the batch back where it was (BatchedAqua.snapshot / restore: bit for bit, tick and
        restart markers included): the timed regions start from the state the warm-up left, and none of them is the
        first launch of its graph.  -> how many graphs were replayed"""
        if not self.use_graph:
            return 0
        seen = set(seg[:3] for n in played for seg in self.plan(n))
        todo = []
        for seg in self.plan(timed_steps):
            if seg[:3] not in seen:
                seen.add(seg[:3])
                todo.append(seg[:3])
        if not todo:
            return 0
        saved = self.env.snapshot()
        for _ in range(max(1, repeats)):
            for key in todo:
                self.graphs[key].launch()
        self.env.restore(saved)                          # (refreshes the device's tick base here, not inside region 0)
        return len(todo)

    def region_is_timed_graph(self, n_steps):
        plan = self.plan(n_steps)
        return self.use_graph and len(plan) == 1 and plan[0][:3] in self.timed

    def region_graph_ms(self, segs):
        """GPU time between the first and the last node of the region's graph (None unless the region was one timed
        graph); call after the stream has been synchronised"""
        if not self.use_graph or len(segs) != 1:
            return None
        key = segs[0][:3]
        if key not in self.timed:
            return None
        try:
            return self.graphs[key].elapsed_ms()
        except Exception as exc:                     # in-graph event records unsupported: the stream events remain
            self.timing_error = "%s: %s" % (type(exc).__name__, exc)
            return None

    def clocked_by_launch_events(self, n_steps):
        """a region of one block can carry events attached to its first and last launch: it is then queued by ONE C call of
        plain launches (hipExtLaunchKernel for those two) instead of a graph, whose nodes cannot carry events.  (The first
        and the last step as launches of their own around the graph of the others was measured too: 5.29 us per step where
        the one call gives 5.14 and graph + stream events 5.46 -- every change of submission path costs.)"""
        return self.launch_events is not None and len(self.plan(n_steps)) == 1

    def prepare_clocked(self, n_steps):
        """a one-block region that carries the launch events: its launches marshalled once (what capture is for a graph)"""
        (buf, row0, s, _), = self.plan(n_steps)
        key = ("clocked", buf, row0, s)
        if key not in self.graphs:
            self.graphs[key] = self.env.prepare_rollout(s, actions=self.actions, keep_all=False,
                                                        done_history=self.hist[buf][row0:row0 + s], events=self.launch_events)

    def run(self, n_steps, after_last_launch=None, before_first_launch=None, clock=False):
        """exactly n_steps steps; returns the segments it queued.  before_first_launch() is called right ahead of the
        region's first step launch, after_last_launch() right behind its last one and ahead of the exchange of its last
        block: where bench.py records its HIP events, so that the exchange's host-side bookkeeping (on an idle stream an
        event is stamped at once) is not inside the interval.  clock=True: a one-block region is queued by rollout() with
        self.launch_events attached to its first and last launch"""
        segs = self.plan(n_steps)
        by_launch = clock and self.clocked_by_launch_events(n_steps)
        handle = self.stream_handle
        for i, (buf, row0, s, gather_after) in enumerate(segs):
            if self.exchange is not None and row0 == 0:
                self.exchange.wait_source(buf)
            # (everything that can be looked up ahead of the region's first event is: on an idle stream the event is stamped
            # at once and the host time until the launch counts as the region's, profiles/r04/launch_gap/)
            graph = (self.graphs[("clocked", buf, row0, s) if by_launch else (buf, row0, s)]
                     if (by_launch or self.use_graph) else None)
            if before_first_launch is not None and i == 0:
                before_first_launch()
            if graph is None:
                self.env.rollout(s, actions=self.actions, keep_all=False, done_history=self.hist[buf][row0:row0 + s])
            elif by_launch or handle is None:
                graph.launch()
            else:
                graph.launch(handle)
            if after_last_launch is not None and i == len(segs) - 1:
                after_last_launch()
            if self.exchange is not None and gather_after:
                self.exchange.gather_async(self.hist[buf], source_id=buf, final=i == len(segs) - 1)
            self.steps_run += s
        if not segs:
            for hook in (before_first_launch, after_last_launch):
                if hook is not None:
                    hook()
        return segs

    @property
    def launch(self):
        return "hipGraph" if self.use_graph else "eager"


def warmup_runs(warmup, steps):
    """The warm-up budget as run lengths.  A captured graph's FIRST launch costs extra (region 0 of round 3's driver line:
    9.3 us per step against 5.5 in the other four), so the warm-up replays the timed region's own graphs where its budget
    allows -- warmup >= steps: whole regions of `steps` steps, then the remainder -- and otherwise the timed graphs are
    uploaded ahead of time (StepRunner.upload_unplayed: hipGraphUpload).  Exactly `warmup` steps either way."""
    if warmup >= steps > 0:
        q, r = divmod(warmup, steps)
        return [steps] * q + ([r] if r else [])
    return [warmup] if warmup else []


def pick_median(values):
    """index of the median element (the lower one of the two middle elements for an even count)"""
    order = sorted(range(len(values)), key=lambda i: values[i])
    return order[(len(values) - 1) // 2]


def region_outliers(us_per_step, factor=OUTLIER_FACTOR):
    """how many regions ran slower than factor x the median, and the slowest one (the median hides a 13-us region)"""
    if not us_per_step:
        return {"n": 0, "factor": factor, "slowest": None}
    med = statistics.median(us_per_step)
    worst = max(range(len(us_per_step)), key=lambda i: us_per_step[i])
    return {"n": sum(1 for v in us_per_step if v > factor * med), "factor": factor,
            "slowest": {"region": worst, "us_per_step": us_per_step[worst], "x_median": us_per_step[worst] / med if med else None}}


def summarize_regions(walls_s, events_ms, steps, n, world, a_bytes):
    """One set of timed regions -> the figures of a line, each fraction beside the rate it follows from (SURVEY.md 8d:
    achieved = A x steps per second): wall clock (MAX over the ranks, barrier-bracketed: `value`) and HIP events on rank
    0's launch stream (`value_by_events`, `frac`)."""
    wall = walls_s[pick_median(walls_s)] / steps                    # s per step
    ev = statistics.median(events_ms) * 1e-3 / steps
    return {"regions": len(walls_s), "wall_us_per_step": wall * 1e6, "event_us_per_step": ev * 1e6,
            "value": world * n / wall, "value_by_events": world * n / ev,
            "frac_by_wall": a_bytes * n / wall / 1e9 / HBM_PEAK_GBPS, "frac": a_bytes * n / ev / 1e9 / HBM_PEAK_GBPS}


class RegionClock(object):
    """Times sets of regions of `steps` steps on this rank.  A region: opening bracket = the closing bracket of what ran
    before it (barrier + synchronize); the clock starts; the runner queues the steps (HIP events recorded on `stream` right
    around its launches); drain() -- this rank's queue is empty, steps and done-mask copies; the clock stops; rendezvous().
    Without a device (torch None: the CPU tests) the wall clock stands in for the events."""

    def __init__(self, runner, steps, torch=None, dist=None, stream=None, settle_us=0.0, on_region=None):
        self.runner, self.steps, self.torch, self.dist, self.stream = runner, steps, torch, dist, stream
        self.settle_us, self.on_region = settle_us, on_region
        self.regions_run = 0

    def drain(self, exchange):
        """everything this rank queued has run: the steps, and the done-mask gathers behind them on the side streams"""
        if exchange is not None:
            exchange.finish()
        if self.torch is not None:
            self.torch.cuda.synchronize()

    def rendezvous(self, exchange):
        """barrier + synchronize: the closing bracket of one region is the opening bracket of the next.  The clock of a
        region stops at this rank's own drain() -- the region's time is the MAX over the ranks of that, which is when the
        slowest rank was done; the barrier's own latency is not part of any region.  It is also the fence of the
        done-mask exchange: every rank has drained, so every block published in the region is in place everywhere"""
        if self.dist is not None:
            self.dist.barrier()
            if self.torch is not None:
                self.torch.cuda.synchronize()
        if exchange is not None:
            exchange.note_fence()

    def run(self, n_regions, exchange):
        """-> (wall seconds per region, event milliseconds per region, the last region's segments); call between two
        rendezvous() (it ends with one)"""
        walls, events, segs = [], [], []
        torch, stream = self.torch, self.stream
        for _ in range(n_regions):
            if torch is not None:
                # HIP events on the stream the step kernels are launched on, around the region's launches (the contract's
                # `roofline` clock).  Round 2 timed a one-graph region with event-record NODES inside the graph instead:
                # measured side by side (profiles/r03/k20_event_methods.txt) the two nodes add ~7 us of their own to a
                # 100-us graph
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                before, after = (lambda: e0.record(stream)), (lambda: e1.record(stream))
            else:
                before = after = None
            if self.settle_us > 0:
                time.sleep(self.settle_us * 1e-6)
            t0 = time.perf_counter()
            segs = self.runner.run(self.steps, before_first_launch=before, after_last_launch=after, clock=True)
            self.drain(exchange)
            walls.append(time.perf_counter() - t0)
            self.rendezvous(exchange)
            events.append(e0.elapsed_time(e1) if torch is not None else walls[-1] * 1e3)          # ms
            if self.on_region is not None:
                self.on_region(segs)
            self.regions_run += 1
        return walls, events, segs


def rolls_back_first_replay(first_replay, use_graph, by_launch, regions):
    """whether the timed graphs the warm-up did not replay are replayed once, untimed, and taken back ahead of region 0:
    only where region 0 could move the reported median (fewer than ROLLBACK_BELOW_REGIONS regions)"""
    return first_replay == "rollback" and use_graph and not by_launch and regions < ROLLBACK_BELOW_REGIONS


def exchange_ab_legs(main_kind, have_exchange):
    """The extra legs behind the main regions of a run with a done-mask exchange (N > 1): the step path alone, then every
    transport the main regions did NOT use -- so that ONE run on a node measures RCCL's all-gather (what BASELINE.json's
    north_star names) even when `auto` settled on IPC peer copies, and separates what the kernels scale like from what the
    exchange costs."""
    if not have_exchange and main_kind is None:
        return ["ipc", "rccl"]                      # (the main regions already are the step path alone)
    return ["step_only"] + [k for k in ("ipc", "rccl") if k != main_kind]


def run_exchange_ab(clock, runner, legs, open_leg, n_regions, summarize, agree, guard=None, check=None, out=None):
    """Time `n_regions` regions per leg, one leg after the other, every rank in step.  open_leg(name) returns the leg's
    exchange (None for "step_only") on EVERY rank or raises on every rank (sharded.open_exchange's own agreement);
    agree(error, what) raises on every rank if any rank passes an error; summarize(walls, events) is collective (MAX of
    the wall clocks over the ranks).  A leg that cannot be set up is recorded and skipped; a leg whose regions or check fail
    ends the legs for every rank together -- neither touches what was measured before.  (A failure on ONE rank in the
    middle of a region leaves the others in that region's barrier: no agreement reaches them, the guard's deadline does.)  `out` is filled as
    the legs complete, so that a deadline (guard: a sharded.Watchdog whose stage is moved along) finds what there is."""
    out = {} if out is None else out
    main = runner.exchange
    try:
        for name in legs:
            if guard is not None:
                guard.stage = "exchange_ab leg '%s': set-up" % name
            try:
                ex = open_leg(name)
            except Exception as exc:
                out[name] = {"error": "set-up: %s: %s" % (type(exc).__name__, exc)}
                continue
            if guard is not None:
                guard.stage = "exchange_ab leg '%s': timed regions" % name
            runner.exchange = ex
            err, walls, events, extra = None, None, None, None
            try:
                walls, events, segs = clock.run(n_regions, ex)
                extra = check(ex, segs) if (check is not None and ex is not None) else None
            except Exception as exc:
                err = exc
            try:
                agree(err, "exchange_ab leg '%s'" % name)
            except RuntimeError as exc:
                out[name] = {"error": "timed regions: %s" % exc}
                out["error"] = "legs stopped in '%s'" % name
                if ex is not None:
                    ex.abandon()
                break
            row = summarize(walls, events)
            if extra:
                row.update(extra)
            out[name] = row
            if ex is not None:
                if guard is not None:
                    guard.stage = "exchange_ab leg '%s': closing its exchange" % name
                ex.close()
    finally:
        runner.exchange = main
    return out


def episodes_ended(hist, segs, np):
    """done flags set in the rows the given segments wrote (popcount of the ballot words)"""
    total = 0
    for buf, row0, s, _ in segs:
        words = hist[buf][row0:row0 + s]
        words = words.cpu().numpy() if hasattr(words, "cpu") else np.asarray(words)
        total += popcount_words(np.ascontiguousarray(words), np)
    return total


def popcount_words(words, np):
    """set bits of an array of 64-bit words (np.bitwise_count needs numpy >= 2)"""
    w = np.ascontiguousarray(words).view(np.uint64)
    if hasattr(np, "bitwise_count"):
        return int(np.bitwise_count(w).sum())
    return int(np.unpackbits(w.view(np.uint8)).sum())


# ------------------------------------------------------------------ CPU baselines (oracle/: the checker, timed beside the GPU)
def _port_worker(job):
    obstacles, continuous, budget_s, seed = job
    from oracle.aqua_oracle import time_scalar_port
    return time_scalar_port(obstacles, continuous, budget_s=budget_s, seed=seed)


def usable_cores():
    """cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows all
    of the host's cores in the mask but schedules the container on its share of them)"""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = fields[0], float(fields[1])
            else:
                quota = fields[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    period = float(g.read().split()[0])
            if quota not in ("max", "-1") and float(quota) > 0:
                cores = max(1, min(cores, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return cores


def host_info():
    """what the CPU baselines ran on: model name (/proc/cpuinfo), logical CPUs the OS shows, cores this process may use"""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "usable_cores": usable_cores()}


def cpu_baselines(args, obstacles):
    """Timed on this host's cores, rank 0 at N = 1 only, bounded samples of the same workload (random actions,
    reset on done, the same obstacle set):
      cpu_baseline       reference-style port (oracle.ScalarPort: one world per Python object, numpy float64, the
                         reference's operation sequence, gym_aqua/envs/aqua.py:135-213) on ALL cores, one process per
                         core with independent worlds -- the loop of main/testing/__init__.py:17-36 under multiprocessing;
      cpu_baseline_1core the same port on one core;
      cpu_baseline_c     the float64 C restatement (oracle/aqua_oracle.c) over 262 144 worlds, OpenMP on all cores.
    Must run before the process initialises HIP (worker processes are forked + exec'd)."""
    import multiprocessing as mp
    import numpy as np
    from oracle.aqua_oracle import COracle, time_scalar_port
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)       # read by libgomp when the C oracle is loaded below
    budget = float(args.cpu_seconds)
    steps1, secs1 = time_scalar_port(obstacles, args.continuous, budget_s=min(budget, 5.0))
    one = {"value": steps1 / secs1, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": "%d random-action steps of ONE world (reset on done), oracle.ScalarPort, %.1f s" % (steps1, secs1)}
    # one process per core; ProcessPoolExecutor so that a worker that cannot start raises instead of hanging
    from concurrent.futures import ProcessPoolExecutor
    try:
        with ProcessPoolExecutor(cores, mp_context=mp.get_context("spawn")) as pool:
            list(pool.map(_port_worker, [(obstacles, args.continuous, 0.05, 100 + i) for i in range(cores)],
                          timeout=300))                                       # workers up, imports done
            t0 = time.perf_counter()
            parts = list(pool.map(_port_worker, [(obstacles, args.continuous, budget, i) for i in range(cores)],
                                  timeout=budget + 300))
            wall = time.perf_counter() - t0
        steps_all = sum(p[0] for p in parts)
        allc = {"value": steps_all / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
                "sample": "%d random-action steps over %d independent worlds, one process per core (multiprocessing), "
                          "oracle.ScalarPort (reset on done), %.1f s wall" % (steps_all, cores, wall)}
    except Exception as exc:                                                  # report the one-core figure, say why
        allc = dict(one, sample=one["sample"] + " [all-cores leg failed: %s: %s]" % (type(exc).__name__, exc))
    orc = COracle()
    n = 262144
    st = np.zeros((7, n), dtype=np.float32)
    tt = np.zeros(n, dtype=np.int32)
    orc.reset(st, tt, obstacles=obstacles, waves=1, seed=0, tick=0)
    orc.rollout_f32(st, tt, 2, obstacles=obstacles, continuous=args.continuous, seed=0, tick0=1)
    t0 = time.perf_counter()
    T, done = 0, 0.0
    while done < min(budget, 8.0):
        orc.rollout_f32(st, tt, 10, obstacles=obstacles, continuous=args.continuous, seed=0, tick0=3 + T)
        T += 10
        done = time.perf_counter() - t0
    c = {"value": n * T / done, "unit": "env-steps/s", "cores": orc.threads(), "kind": "port",
         "sample": "%d steps of %d worlds, C float64 oracle (oracle/aqua_oracle.c) with OpenMP, %.1f s" % (T, n, done)}
    return allc, one, c


# ------------------------------------------------------------------ self-launch of the N > 1 form (tests/test_bench_logic.py)
def needs_self_launch(gpus, environ):
    """`bench.py --gpus N` with N > 1 was started bare (no launcher set WORLD_SIZE / RANK): it must start the ranks itself"""
    return gpus > 1 and "RANK" not in environ and int(environ.get("WORLD_SIZE", "1")) == 1


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_command(gpus, argv, port, python=None, script=None):
    """the driver's own N > 1 command line around the arguments this process was given"""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


FALLBACK_DEADLINE_S = 150.0


def fallback_chain(exchange, one_gpu):
    """transports bench.py's self-launch tries in turn, each with FRESH ranks: only `auto` falls back"""
    if exchange != "auto":
        return [exchange]
    return ["auto", "none"] if one_gpu else ["auto", "rccl", "none"]


def argv_with(argv, exchange, note):
    """argv with --exchange / --exchange-note replaced"""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in ("--exchange", "--exchange-note"):
            skip = True
            continue
        if a.startswith("--exchange=") or a.startswith("--exchange-note="):
            continue
        out.append(a)
    out += ["--exchange", exchange]
    if note:
        out += ["--exchange-note", note]
    return out


def run_ranks(cmd, env, deadline_s, out=None, err=None):
    """Start `cmd` as a fresh child in a session of its own and wait at most deadline_s for it.
    -> (status, first JSON line or None, timed out).  Everything else the child prints on stdout goes to stderr.  Past the
    deadline the child's whole process group (the launcher and its ranks: the group it leads, by number) is killed."""
    import signal
    import subprocess
    import threading
    out, err = out or sys.stdout, err or sys.stderr
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    lines = []

    def pump():
        for line in proc.stdout:                        # rank 0's line goes to stdout, everything else to stderr
            if line.startswith("{") and not lines:
                lines.append(line)
            else:
                err.write(line)
                err.flush()
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    timed_out = False
    try:
        status = proc.wait(timeout=deadline_s)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)                # the session we created: its leader's pid IS the group's number
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        status = proc.wait()
    t.join(timeout=5)
    return status, (lines[0] if lines else None), timed_out


def self_launch(args, argv, device_count=None, run=None):
    """Start the ranks as a FRESH child process (never a re-exec: this process may not replace itself once anything has
    touched the GPU, and the children must initialise HIP themselves), relay rank 0's JSON line, return the children's
    status.  Refuses, with a message and a non-zero status, when the box has fewer GPUs than ranks.
    With --exchange auto a set of ranks that exits non-zero, or is still running at its deadline, is followed ONCE by fresh
    ranks with --exchange rccl and then ONCE with --exchange none (the step path alone; the line then carries
    config.done_mask_exchange false and the failures in done_mask_exchange_note): the first line that arrives is relayed
    and its ranks' status returned -- first contact with a node cannot end without a line because a transport misbehaved."""
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count()       # does not initialise HIP on this image
    if device_count < args.gpus and not args.ranks_on_one_gpu:
        sys.stderr.write("bench.py --gpus %d: this box shows %d GPU(s); one rank per GPU is the only supported layout "
                         "(--ranks-on-one-gpu rehearses the path on one device, without meaningful timings)\n"
                         % (args.gpus, device_count))
        return 2
    if device_count < 1:
        sys.stderr.write("bench.py: no GPU visible\n")
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    run = run or run_ranks
    failures, status = [], 1
    chain = fallback_chain(args.exchange, args.ranks_on_one_gpu)
    for i, exchange in enumerate(chain):
        note = "; ".join(failures) or args.exchange_note
        cmd = launch_command(args.gpus, argv_with(argv, exchange, note) if (i or note) else argv, free_port())
        deadline = args.launch_deadline if i == 0 else min(args.launch_deadline, FALLBACK_DEADLINE_S)
        status, line, timed_out = run(cmd, env, deadline)
        if line is not None:
            sys.stdout.write(line)
            sys.stdout.flush()
            return status
        failures.append("--exchange %s: %s" % (exchange, ("no line after %g s (ranks killed)" % deadline) if timed_out
                                                else "ranks exited with status %d and no line" % status))
        sys.stderr.write("bench.py: %s%s\n" % (failures[-1], "; starting fresh ranks with --exchange %s" % chain[i + 1]
                                                if i + 1 < len(chain) else ""))
    return status or 1


# ------------------------------------------------------------------ the done-mask exchange of this run
def exchange_blocks(n_steps, chunk, block_rows):
    """done-mask blocks a region of n_steps publishes (kind="ipc" needs that many receive slots between two fences)"""
    return sum(1 for seg in plan_region(n_steps, chunk, block_rows) if seg[3])


# ------------------------------------------------------------------ main
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if needs_self_launch(args.gpus, os.environ):
        sys.exit(self_launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        args.gpus = world

    from aquaticgymenv_amd import presets
    obstacles = presets.NONE if args.no_obstacles else presets.BENCH8
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baselines(args, obstacles)          # before anything touches the GPU (see the docstring)

    # rank 0's ONE JSON line is the only thing this process may put on stdout: RCCL prints a version banner there when its
    # communicator comes up, other libraries may chat too.  From here on file descriptor 1 is stderr; the line goes to
    # a private duplicate of the original stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from aquaticgymenv_amd.batched import BatchedAqua, LaunchEvents
    from aquaticgymenv_amd.sharded import Watchdog, open_exchange

    one_gpu = args.ranks_on_one_gpu
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL, mapped buffers): before the first HIP call
    dev = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(dev)
    distributed = world > 1 or "RANK" in os.environ
    # Everything from here to the end of the exchange's set-up can STALL rather than fail when a node is met for the first
    # time (a rendezvous, an IPC mapping, a first copy over xGMI): the rendezvous has its own timeout, and a watchdog turns
    # whatever else hangs into one stderr line + exit status 3 within --setup-deadline (cancelled once the set-up is done)
    watchdog = Watchdog(args.rendezvous_timeout + args.setup_deadline, "rendezvous (init_process_group)") if distributed else None
    if distributed:
        from datetime import timedelta
        limit = timedelta(seconds=args.rendezvous_timeout)
        if one_gpu:
            dist.init_process_group("gloo", timeout=limit)   # RCCL refuses two ranks on one device; gloo carries the barriers
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        watchdog.stage = "first collective (all_gather_object of the device names)"
    ranks_seen = dist.get_world_size() if distributed else 1
    if distributed:
        devices = [None] * ranks_seen
        dist.all_gather_object(devices, "rank %d: cuda:%d %s" % (rank, dev.index, torch.cuda.get_device_name(dev)))
    else:
        devices = ["rank 0: cuda:%d %s" % (dev.index, torch.cuda.get_device_name(dev))]
    launch_stream = torch.cuda.Stream(device=dev)        # the step kernels' stream; the HIP events are recorded on it
    torch.cuda.set_stream(launch_stream)

    n = args.envs
    env = BatchedAqua(n, obstacles=obstacles, continuous=args.continuous, seed=0, env_offset=rank * n,
                      auto_reset=0 if args.no_auto_reset else args.reset_mode, device=dev)
    env.reset()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    chunk = min(CHUNK, max(args.steps, 1))
    if args.continuous:
        actions = torch.rand((CHUNK, 2, env.ld), device=dev, generator=gen) * 0.3 + 0.2
    else:
        actions = torch.randint(0, 3, (CHUNK, env.ld), device=dev, generator=gen, dtype=torch.int64).to(torch.uint8)
    words = env.ld // 64
    # rows of one done-mask block = one all-gather: at most GATHER_EVERY chunks, and no more than a region holds (a
    # 20-step region must not ship a 500-row buffer through RCCL)
    block_rows = max(chunk, min(GATHER_EVERY * CHUNK, args.steps))
    hist = [torch.zeros((block_rows, words), dtype=torch.int64, device=dev) for _ in range(2)]
    exchange, exchange_kind, exchange_note = None, None, args.exchange_note
    slots = max([exchange_blocks(w, chunk, block_rows) for w in warmup_runs(args.warmup, args.steps)] +
                [exchange_blocks(args.steps, chunk, block_rows), 1])
    if world > 1 or args.force_exchange:
        if watchdog is not None:
            watchdog.stage = "setting up the done-mask exchange (--exchange %s)" % args.exchange
        # no failure and no stall of a transport ends the run: ipc -> rccl -> none, decided by all ranks together
        # (sharded.open_exchange); with the ranks on one GPU there is no RCCL to fall back to (gloo carries the barriers)
        exchange, exchange_kind, note = open_exchange(args.exchange, block_rows, words, dev, slots=slots,
                                                      copy_engine=args.copy_engine, soft_deadline_s=args.soft_deadline,
                                                      allow_rccl=not one_gpu)
        exchange_note = "; ".join(n for n in (args.exchange_note, note) if n) or None
    if watchdog is not None:
        watchdog.cancel()
    runner = StepRunner(env, actions, hist, exchange, use_graph=not args.eager, chunk=chunk,
                        launch_events=LaunchEvents() if args.region_clock == "launch" else None)
    runner.set_stream(launch_stream)                     # (torch's current stream since set_stream() above)
    by_launch = runner.clocked_by_launch_events(args.steps)
    if args.region_clock == "launch" and not by_launch:
        raise SystemExit("--region-clock launch: a region of %d steps is more than one block of %d" % (args.steps, chunk))
    warm = warmup_runs(args.warmup, args.steps)
    for w in warm:
        runner.prepare(w)
    if by_launch:
        runner.prepare_clocked(args.steps)
    else:
        runner.prepare(args.steps, timing=args.graph_node_events)
    # region 0 must not be the first launch of its graphs (see warmup_runs)
    uploaded = 0 if by_launch else runner.upload_unplayed(args.steps, warm)
    first_replay = ("no graph" if (by_launch or not runner.use_graph) else
                    "warm-up" if not uploaded else "hipGraphUpload (%d graph%s the warm-up does not replay)" % (uploaded, "s" * (uploaded != 1)))

    node_ms, launch_ev = [], []

    def diagnostics(segs):
        launch_ev.append(runner.launch_events.elapsed_ms() if by_launch else None)
        node_ms.append(runner.region_graph_ms(segs) if args.graph_node_events and not by_launch else None)

    clock = RegionClock(runner, args.steps, torch=torch, dist=dist if distributed else None, stream=launch_stream,
                        settle_us=args.settle_us, on_region=diagnostics)
    for i, w in enumerate(warm):
        if i:
            clock.rendezvous(exchange)                   # (every run is a window of the exchange)
        runner.run(w)
        clock.drain(exchange)
    clock.drain(exchange)
    x_before = float(env.state[0, :n].double().sum().item())
    # A timed graph the warm-up never replayed: with few regions its first launch would BE the sample (region 0 of round
    # 3's driver line: 9.3 us per step against 5.5), so it is replayed once, untimed, and the batch put back; with
    # ROLLBACK_BELOW_REGIONS regions or more region 0 cannot move the median and nothing runs beyond the W warm-up steps
    # (hipGraphUpload above is a set-up call, not steps).  Either way the line says so: config.untimed_steps_beyond_warmup
    untimed_beyond_warmup = 0
    if rolls_back_first_replay(args.first_replay, runner.use_graph, by_launch, args.regions):
        rolled_back = runner.replay_unplayed_and_roll_back(args.steps, warm, repeats=args.ramp_replays)
        torch.cuda.synchronize()
        if rolled_back:
            untimed_beyond_warmup = args.steps * max(1, args.ramp_replays)
            first_replay = ("one untimed replay rolled back to the state after the warm-up (%d graph%s the warm-up does not replay)"
                            % (rolled_back, "s" * (rolled_back != 1)))
    clock.rendezvous(exchange)
    walls, events, segs = clock.run(args.regions, exchange)
    my_events = list(events)

    def max_over_ranks(values):
        """the region's time is when the slowest rank was done"""
        if not distributed:
            return list(values)
        tmax = torch.tensor(values, dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return [float(v) for v in tmax.cpu()]

    walls = max_over_ranks(walls)
    per_rank = None
    if distributed:
        # every rank's own clocks: one slow GPU (or one slow link) shows here, not in a MAX
        per_rank = [None] * ranks_seen
        dist.all_gather_object(per_rank, {"rank": rank, "device": devices[rank] if rank < len(devices) else None,
                                          "event_us_per_step": statistics.median(my_events) * 1e3 / args.steps,
                                          "event_us_per_step_max": max(my_events) * 1e3 / args.steps})
    timed_steps = args.regions * args.steps

    # sanity on the timed work (no cached / skipped work): every step was queued, the worlds moved, and -- with
    # restarts on -- episodes ended inside the last timed region (reported, not asserted: a one-step region right
    # after a reset may legitimately end none)
    expect = args.warmup + timed_steps
    if env._tick != expect or runner.steps_run != expect:
        raise RuntimeError("bench queued %d steps (env tick %d), expected %d" % (runner.steps_run, env._tick, expect))
    x_after = float(env.state[0, :n].double().sum().item())
    if not (x_after != x_before):
        raise RuntimeError("the worlds did not move during the timed regions (x checksum %r -> %r)" % (x_before, x_after))
    ended = episodes_ended(hist, segs, np)
    exchange_check = None
    if exchange is not None and segs:
        # the last block as it arrived: this rank's own copy must BE its done-mask buffer, and the other ranks' blocks must
        # show episodes ending there too (after the closing barrier every block of the region is in place everywhere)
        got = exchange.gathered[exchange.last_slot()]
        peers = [int(popcount_words(got[r].cpu().numpy(), np)) for r in range(ranks_seen) if r != rank]
        exchange_check = {"own_block_intact": bool(torch.equal(got[rank], hist[segs[-1][0]])),
                          "episodes_in_peer_blocks": peers}
        if not exchange_check["own_block_intact"]:
            raise RuntimeError("rank %d: the done-mask block it received from itself differs from the block it sent" % rank)

    result = None
    if rank == 0:
        a_bytes = A_CONTINUOUS if args.continuous else A_DISCRETE
        mid = pick_median(walls)
        wall = walls[mid]
        steps_per_s = world * n * args.steps / wall
        # the step kernel is the only kernel in the timed stream: its average launch period on the launch stream
        # (HIP events around the region: inter-kernel boundaries and the event-to-first-kernel gap included, so
        # this is an upper bound of the kernel's own duration); median over the regions
        # (a one-block region carries its events ON its first and last launch instead: the two stream events are marker
        # packets of their own and read 12-14 us more per region than first kernel start -> last kernel end)
        launch_s = statistics.median(launch_ev if by_launch else events) * 1e-3 / args.steps
        achieved = a_bytes * n / launch_s / 1e9
        first_s = statistics.median((launch_ev if by_launch else events)[:REGIONS]) * 1e-3 / args.steps
        first_wall = statistics.median(walls[:REGIONS])
        traffic, traffic_src = committed_traffic(n, args)
        # `value` counts every world of the batch in every step.  With next-step restarts a world that finished at tick t
        # does not step during tick t + 1 (it is re-seeded and reports reward 0, term 0): those ticks are the episodes that
        # ended one tick earlier -- counted here from the last region's own done masks (rank 0's shard)
        restart_ticks = ended if (not args.no_auto_reset and args.reset_mode == 2) else 0
        restart_frac = restart_ticks / float(max(args.steps * n, 1))
        issue_us, issue_src = issue_bound(n, args, ended / float(max(args.steps, 1)))
        result = {
            "metric": "env-steps/sec at batch=262144; achieved HBM GB/s vs roofline; 1/2/4/8-GPU scaling",
            "value": steps_per_s, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the two clocks side by side, each rate beside the roofline fraction that follows from it (SURVEY.md 8d:
            # achieved = A x steps per second): value <-> roofline.frac_by_wall (wall clock, MAX over the ranks, barriers
            # between the regions); value_by_events <-> roofline.frac (HIP events on rank 0's launch stream)
            "value_by_events": world * n / launch_s,
            "regions": args.regions, "regions_rule": args.regions_rule, "regions_ms": [w * 1e3 for w in walls],
            "region_reported": "median",
            "regions_outliers": {"by_events": region_outliers([e * 1e3 / args.steps for e in (launch_ev if by_launch else events)]),
                                 "by_wall": region_outliers([w * 1e6 / args.steps for w in walls])},
            "config": {"workload": "batch %d worlds/GPU, %s actions, %s, waves on, auto-reset (%s), %s of <= %d steps"
                       % (n, "continuous f32x2" if args.continuous else "discrete u8",
                          "no obstacles" if args.no_obstacles else "4 circle + 4 rect obstacles",
                          "off" if args.no_auto_reset else ("next-step" if args.reset_mode == 2 else "same-step"),
                          "one C call of launches per region" if by_launch else
                          ("HIP graphs" if runner.use_graph else "eager launch loops"), chunk),
                       "baseline_config": "configs[3]" if args.continuous else ("configs[1]-like" if args.no_obstacles else "configs[2]"),
                       "worlds_per_gpu": n, "global_worlds": world * n, "parallelism": "range-partition x%d" % world,
                       "launch": "eager" if by_launch else runner.launch, "timed_graph_first_replay": first_replay,
                       "untimed_steps_beyond_warmup": untimed_beyond_warmup,
                       "done_mask_exchange": exchange is not None,
                       "done_mask_exchange_kind": exchange_kind, "done_mask_exchange_note": exchange_note,
                       "done_mask_copy_engine": args.copy_engine if exchange_kind == "ipc" else None,
                       "ranks_seen": ranks_seen, "devices": devices,
                       "process_group": (("gloo (ranks share cuda:0: rehearsal, timings meaningless)" if one_gpu else "nccl (RCCL)")
                                         if distributed else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "frac_of_measured_copy_peak": achieved / HBM_COPY_GBPS,
                         # ... on live world-steps only (a restart tick moves no boat: sanity.restart_ticks_fraction)
                         "frac_live": achieved / HBM_PEAK_GBPS * (1.0 - restart_frac),
                         "frac_by_wall": a_bytes * n / (wall / args.steps) / 1e9 / HBM_PEAK_GBPS,
                         # the protocol-independent figure, the one to compare across rounds: the kernel's own average
                         # duration under rocprofv3 --kernel-trace with this command's flags, committed for THIS build
                         "kernel_only": kernel_only(n, args, a_bytes),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_world_step": a_bytes, "launch_us": launch_s * 1e6,
                         "issue_bound_us": issue_us, "issue_bound_source": issue_src,
                         "first_regions": {"n": min(REGIONS, args.regions), "launch_us": first_s * 1e6,
                                           "frac": a_bytes * n / first_s / 1e9 / HBM_PEAK_GBPS,
                                           "ms_per_step": first_wall * 1e3 / args.steps,
                                           "value": world * n * args.steps / first_wall,
                                           "what": "the same figures from the median of the FIRST %d regions alone" % min(REGIONS, args.regions)},
                         "launch_us_regions": [e * 1e3 / args.steps for e in (launch_ev if by_launch else events)],
                         "launch_us_events": "launch" if by_launch else "stream",
                         "launch_us_stream_events_regions": [e * 1e3 / args.steps for e in events],
                         "frac_by_stream_events": a_bytes * n / (statistics.median(events) * 1e-3 / args.steps) / 1e9 / HBM_PEAK_GBPS,
                         "launch_us_graph_nodes_regions": ([g * 1e3 / args.steps if g else None for g in node_ms]
                                                           if args.graph_node_events else None),
                         "note": ("launch_us = HIP-event time of a timed region / its launches, median region, inter-kernel "
                                  "boundaries included.  The two events are ATTACHED to the region's first launch (its start time) "
                                  "and last launch (its end time) -- hipExtLaunchKernel's start/stop events, on the launch stream; the "
                                  "region is one C call of plain launches (a graph node cannot carry events); "
                                  "launch_us_stream_events_regions / frac_by_stream_events: the same regions by two events RECORDED "
                                  "on that stream around them (rounds 1-2's clock: two marker packets, 12-14 us per region more, "
                                  "profiles/r03/burst_timeline.txt).  Kernel-only duration: profiles/" if by_launch else
                                  "launch_us = HIP-event time of a timed region / its launches, median region, inter-kernel "
                                  "boundaries included; the events are recorded on the launch stream around the region's "
                                  "graph launches.  Kernel-only duration: profiles/")},
            "sanity": {"steps_queued": runner.steps_run, "episodes_ended_last_region": ended,
                       "restart_ticks_fraction": restart_frac, "live_world_steps_per_s": steps_per_s * (1.0 - restart_frac),
                       "live_world_steps_per_s_by_events": world * n / launch_s * (1.0 - restart_frac),
                       "done_mask_exchange_last_block": exchange_check},
        }
        if cpu is not None:
            result["cpu_baseline"], result["cpu_baseline_1core"], result["cpu_baseline_c"] = cpu
            result["host"] = host_info()
        if per_rank is not None:
            result["per_rank"] = per_rank
        if args.extras and world == 1:
            result["extras"] = extras(env, torch, n, a_bytes)
        if args.per_world_tables and world == 1:
            result.setdefault("extras", {})["per_world_tables"] = per_world_tables(torch, np, presets, BatchedAqua, n, dev)

    # ---- the line goes out exactly once, whatever the optional legs below do
    import threading
    emit_lock, emitted = threading.Lock(), []

    def emit():
        with emit_lock:
            if rank == 0 and not emitted:
                emitted.append(True)
                result_out.write(json.dumps(result) + "\n")
                result_out.flush()

    # ---- N > 1: the step path alone and the OTHER transport, behind the main regions (never instead of them)
    if (exchange is not None or world > 1) and not args.no_exchange_ab:
        a_bytes = A_CONTINUOUS if args.continuous else A_DISCRETE
        ab = {"regions_per_leg": args.ab_regions, "main": exchange_kind or "none",
              "what": "regions of the same K steps behind the main ones: step_only = no exchange at all; ipc / rccl = the "
                      "transport the main regions did not use (the main one's figures are the line's own, copied here)"}
        if rank == 0:
            result.setdefault("extras", {})["exchange_ab"] = ab
            ab[exchange_kind or "none"] = dict(summarize_regions(walls, events, args.steps, n, world, a_bytes),
                                               own_block_intact=(exchange_check or {}).get("own_block_intact"),
                                               episodes_in_peer_blocks=(exchange_check or {}).get("episodes_in_peer_blocks"))

        def expired(status):
            # an optional leg stalled: the main line is safe -- rank 0 prints it with what the legs gave, everybody leaves
            ab["error"] = "deadline of %g s passed in stage: %s" % (args.ab_deadline, guard.stage)
            emit()
            os._exit(status)

        guard = Watchdog(args.ab_deadline, "exchange_ab", on_expire=expired, status=0)

        def open_leg(name):
            from aquaticgymenv_amd.sharded import injected_faults
            if "ab-stall" in injected_faults() and rank == ranks_seen - 1:
                time.sleep(1.0e6)                       # (tests: a leg's set-up that never returns; --ab-deadline ends it)
            if name == "step_only":
                return None
            if name == "rccl" and one_gpu:
                raise RuntimeError("the ranks share one GPU and gloo carries their barriers: RCCL refuses two ranks on one device")
            if name == "ipc" and exchange_note and "ipc unavailable" in exchange_note:
                raise RuntimeError("not tried again: " + exchange_note)
            ex, _, _ = open_exchange(name, block_rows, words, dev, slots=slots, copy_engine=args.copy_engine,
                                     soft_deadline_s=min(args.soft_deadline, args.ab_deadline / 3.0), allow_rccl=not one_gpu)
            return ex

        def check_leg(ex, leg_segs):
            got = ex.gathered[ex.last_slot()]
            return {"own_block_intact": bool(torch.equal(got[rank], hist[leg_segs[-1][0]])),
                    "episodes_in_peer_blocks": [int(popcount_words(got[r].cpu().numpy(), np)) for r in range(ranks_seen) if r != rank]}

        def summarize_leg(leg_walls, leg_events):
            return summarize_regions(max_over_ranks(leg_walls), leg_events, args.steps, n, world, a_bytes)

        def agree_leg(error, what):
            from aquaticgymenv_amd.sharded import agree
            agree(dist if distributed else None, None, ranks_seen, error, what)

        try:
            run_exchange_ab(clock, runner, exchange_ab_legs(exchange_kind, exchange is not None), open_leg, args.ab_regions,
                            summarize_leg, agree_leg, guard=guard, check=check_leg, out=ab)
        except Exception as exc:                        # a bug in the legs' own bookkeeping must not cost the line either
            ab["error"] = "%s: %s (stage: %s)" % (type(exc).__name__, exc, guard.stage)
        guard.cancel()
        if env._tick != runner.steps_run:               # (the legs' bookkeeping, reported: the main line was checked above)
            ab["error"] = "the legs queued %d steps in all, the batch counted %d" % (runner.steps_run, env._tick)
    emit()
    # The line is out.  What follows is collective tear-down (the exchange's closing fence, a barrier, the process groups):
    # if a rank is gone or out of step by now -- an extra leg that failed on one rank only -- it would wait for ever, and a
    # launcher that never returns can cost the line after all.  Bounded: past the deadline every rank still here leaves, status 0
    teardown = Watchdog(args.teardown_deadline, "tear-down after the line", status=0) if distributed else None
    if exchange is not None:
        exchange.close()                                 # collective: unmaps the peers' buffers behind a fence, ends the pump
    if distributed:
        dist.barrier()
        from aquaticgymenv_amd import sharded
        if any(th.is_alive() for th in sharded.ABANDONED_SETUP_THREADS):
            # a set-up thread of this rank is still inside a call that never returned (the soft deadline left it behind): the
            # line is out, every rank is past the barrier -- end here rather than tear the process groups down around it
            sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()
        teardown.cancel()
    return result


def library_tag():
    """sha256 (first 16 hex digits) of the loaded libaqua_hip.so: ties a committed profile to a build"""
    import hashlib
    from aquaticgymenv_amd import _capi
    try:
        with open(_capi.LIB_PATH, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except Exception:
        return None


def committed_traffic(n, args):
    """(HBM bytes per launch, source) from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate passes, KB -> bytes, FETCH_SIZE doubled: on gfx950 it reports half of a coalesced
    stream, MI355X_MICROARCH.md "HBM").  The counters cannot be read from inside this process, so the figure is
    a committed measurement of the SAME build: (None, reason) when no row matches this configuration or the row's
    build tag is not the loaded library's."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None, "no profiles/traffic.json"
    key = "n%d_%s_%s" % (n, "cont" if args.continuous else "disc", "k0" if args.no_obstacles else "k8")
    row = table.get(key)
    if not row:
        return None, "no committed PMC pass for " + key
    tag = library_tag()
    if row.get("library_sha16") != tag:
        return None, "committed PMC pass %s is of build %s, this is %s" % (row.get("source"), row.get("library_sha16"), tag)
    return (2.0 * row["FETCH_SIZE_per_launch_raw"] * 1024.0 + row["WRITE_SIZE_per_launch_raw"] * 1024.0,
            "%s (build %s)" % (row.get("source"), tag))


def kernel_only(n, args, a_bytes):
    """The benchmarked kernel's own average duration under `rocprofv3 --kernel-trace --stats` (no launch gaps, no event
    markers, no dependence on how many regions bench.py times) from the summary committed under profiles/ for THIS build
    (the kernel-trace row of profiles/traffic.json), as a roofline fraction: the figure to compare from round to round.
    The reason instead when there is no row for this configuration or the row is of another build."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return {"avg_ns": None, "frac": None, "source": "no profiles/traffic.json"}
    key = "n%d_%s_%s" % (n, "cont" if args.continuous else "disc", "k0" if args.no_obstacles else "k8")
    row = table.get("%s_steps%d" % (key, args.steps)) or table.get(key)          # (a trace of this very command, if there is one)
    if not row or "step_kernel_avg_ns" not in row:
        return {"avg_ns": None, "frac": None, "source": "no committed kernel trace for " + key}
    tag = library_tag()
    if row.get("library_sha16") != tag:
        return {"avg_ns": None, "frac": None,
                "source": "committed trace %s is of build %s, this is %s" % (row.get("source"), row.get("library_sha16"), tag)}
    ns = float(row["step_kernel_avg_ns"])
    return {"avg_ns": ns, "calls": row.get("step_kernel_calls"), "kernel": row.get("step_kernel_name"),
            "frac": a_bytes * n / ns / HBM_PEAK_GBPS, "source": "%s (build %s)" % (row.get("source"), tag)}


SHADER_CLOCK_GHZ = 2.4       # measured on these kernels: profiles/r03/clock_probe.txt (2.38-2.43)
SIMDS = 256 * 4              # MI355X: 256 compute units of four SIMDs (MI355X_MICROARCH.md)


def issue_bound(n, args, restarts_per_step):
    """(us per launch, source): the vector-issue floor of the benchmarked kernel from the committed instruction budget of
    the SAME build (profiles/isa_budget.json, written by tools/isa_budget.py from the compiler's own listing): VALU
    issue cycles of a stepping wavefront x the wavefronts a SIMD steps + those of a re-seeding pass x the passes the
    measured restarts need, / the shader clock.  A launch cannot be shorter than this however its memory traffic goes; the
    HBM figure (`achieved`, `frac`) is the contract's, this is what actually bounds a 262 144-world launch."""
    path = os.path.join(ROOT, "profiles", "isa_budget.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None, "no profiles/isa_budget.json"
    key = "ns_%s_%s" % ("f32x2" if args.continuous else "u8", "k0" if args.no_obstacles else "k8")
    row = table.get(key)
    if not row or args.no_auto_reset or args.reset_mode != 2:
        return None, "no committed instruction budget for this configuration"
    tag = library_tag()
    if row.get("library_sha16") != tag:
        return None, "committed budget is of build %s, this is %s" % (row.get("library_sha16"), tag)
    # (a re-seeding pass of a wavefront serves up to eight worlds: restarts / 8 passes at the least)
    cycles = (n / 64.0) * row["valu_cycles_stepping_wavefront"] + (restarts_per_step / 8.0) * row["valu_cycles_reseed_pass"]
    return cycles / SIMDS / (SHADER_CLOCK_GHZ * 1e3), "%s (build %s)" % (row.get("source"), tag)


def per_world_tables(torch, np, presets, BatchedAqua, n, dev):
    """separate lines: every world with its own obstacle list (BENCH8, each obstacle moved by up to +-3 units per
    world), HIP graph of 100 steps, both restart modes.  Algorithmic bytes per world-step: 62 + 24 per obstacle row
    read (6 float32 per row)."""
    rng = np.random.RandomState(7)
    tables = np.repeat(presets.BENCH8[None], n, axis=0).astype(np.float64)
    tables[:, :, 0:2] += rng.uniform(-3, 3, (n, 8, 2))
    a_bytes = A_DISCRETE + 24 * 8
    out = {"algorithmic_bytes_per_world_step": a_bytes, "layout": "per-world obstacle tables [K][6][N] float32"}
    for mode in ("next_step", "same_step"):
        env = BatchedAqua(n, obstacles=tables, seed=0, auto_reset=mode, device=dev)
        env.reset()
        g = torch.Generator(device=dev).manual_seed(99)
        actions = torch.randint(0, 3, (CHUNK, env.ld), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
        graph = env.capture_rollout(CHUNK, actions=actions, keep_all=False)
        for _ in range(3):
            graph.launch()
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * CHUNK)
        out[mode] = {"env_steps_per_s": n / us * 1e6, "us_per_step": us, "launches_per_step": 1,
                     "achieved_GBps": a_bytes * n / us / 1e3, "frac_of_8TBps": a_bytes * n / us / 1e3 / HBM_PEAK_GBPS}
        # the fused per-world rollout (state in registers, tables in LDS; 5 B per world-step): its own line, never `value`
        fused = env.capture_rollout(CHUNK, actions=actions, fused=True, keep_all=False)
        for _ in range(3):
            fused.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fused.launch()
        e1.record()
        torch.cuda.synchronize()
        fus = e0.elapsed_time(e1) * 1e3 / (reps * CHUNK)
        out[mode + "_fused_rollout"] = {"env_steps_per_s": n / fus * 1e6, "us_per_step": fus, "steps_per_launch": CHUNK,
                                        "bytes_per_world_step": 5}
        del graph, fused, env
    return out


def extras(env, torch, n, a_bytes):
    """separate lines that must not be mixed into `value`: the fused multi-step kernel (state in
    registers: 5 B per world-step: reward 4 + term 1, actions sampled on the device)."""
    out = {}
    T = 2000
    env.rollout(T, fused=True, keep_all=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.rollout(T, fused=True, keep_all=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    out["fused_rollout"] = {"env_steps_per_s": n * T / (ms * 1e-3), "steps": T, "bytes_per_world_step": 5,
                            "us_per_step": ms * 1e3 / T, "actions": "sampled on device",
                            "note": "ONE launch for all steps, state in registers: the issue-bound floor of this arithmetic"}
    return out


if __name__ == "__main__":
    main()
