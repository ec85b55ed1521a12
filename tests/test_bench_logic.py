"""bench.py's host logic on the CPU: the step queue (chunks, remainder, done-mask blocks, exchange hand-offs) with a
stub environment -- any --steps >= 1 / --warmup >= 0 must queue exactly that many steps, write the done masks of
every one of them, and label the launch mode by what was actually timed.  (Round 1's bench crashed on the driver's
`--steps 20 --warmup 5`: the remainder path wrote no done history.)"""
import numpy as np
import pytest

import bench


class StubGraph(object):
    def __init__(self, env, steps, done_history, how="graph"):
        self.env, self.steps, self.done_history, self.how = env, steps, done_history, how
        self.launches = self.uploads = 0

    def launch(self, stream=None):
        self.launches += 1
        self.streams = getattr(self, "streams", []) + [stream]
        self.env._do(self.steps, self.done_history, self.how)

    def upload(self):
        assert self.launches == 0, "an upload is for a graph that has not been launched yet"
        self.uploads += 1


class StubEnv(object):
    """records what bench.StepRunner asks for; every step sets one bit in its done-mask row"""

    def __init__(self):
        self._tick = 0
        self.captured = []
        self.calls = []

    def _do(self, steps, done_history, how):
        assert done_history is not None and done_history.shape[0] == steps, "every step needs a done-mask row"
        for i in range(steps):
            done_history[i, 0] = self._tick + i + 1          # nonzero: episodes "ended"
        self._tick += steps
        self.calls.append((how, steps))

    def snapshot(self):
        self.calls.append(("snapshot", self._tick))
        return {"_tick": self._tick}

    def restore(self, snap):
        self.calls.append(("restore", snap["_tick"]))
        self._tick = snap["_tick"]

    def capture_rollout(self, steps, actions=None, keep_all=False, done_history=None, timing=False):
        assert steps >= 1
        self.captured.append(steps)
        return StubGraph(self, steps, done_history)

    def rollout(self, steps, actions=None, keep_all=False, done_history=None, events=None):
        self._do(steps, done_history, "eager" if events is None else "eager+events")

    def prepare_rollout(self, steps, actions=None, keep_all=False, done_history=None, events=None):
        assert events is not None
        self.prepared = getattr(self, "prepared", []) + [steps]
        return StubGraph(self, steps, done_history, how="prepared+events")


class StubExchange(object):
    def __init__(self):
        self.log = []

    def wait_source(self, buf):
        self.log.append(("wait", buf))

    def gather_async(self, block, source_id=None, final=False):
        self.log.append(("gather", source_id))
        self.finals = getattr(self, "finals", []) + [final]


def _hist(rows=bench.CHUNK * bench.GATHER_EVERY, words=4):
    return [np.zeros((rows, words), dtype=np.int64) for _ in range(2)]


@pytest.mark.parametrize("n", [0, 1, 5, 20, 99, 100, 101, 250, 499, 500, 501, 1234, 2000])
def test_plan_covers_exactly_n_steps(n):
    segs = bench.plan_region(n)
    assert sum(s for _, _, s, _ in segs) == n
    rows = bench.CHUNK * bench.GATHER_EVERY
    seen = {}
    for buf, row0, s, gather in segs:
        assert 1 <= s <= bench.CHUNK and 0 <= row0 and row0 + s <= rows and buf in (0, 1)
        assert gather == (row0 + s == rows or (buf, row0, s, gather) == segs[-1])
    # rows of one block are written once per pass over it, in order
    buf, row = 0, 0
    for b, row0, s, _ in segs:
        assert (b, row0) == (buf, row)
        row += s
        if row == rows:
            buf, row = buf ^ 1, 0
    if n:
        assert segs[-1][3], "the last segment of a region is always exchanged"
    del seen


def test_plan_rejects_nonsense():
    with pytest.raises(ValueError):
        bench.plan_region(-1)
    with pytest.raises(ValueError):
        bench.plan_region(10, chunk=0)


@pytest.mark.parametrize("steps,warmup", [(20, 5), (1, 0), (100, 100), (101, 7), (2000, 200), (37, 0), (600, 1)])
@pytest.mark.parametrize("use_graph", [True, False])
def test_runner_queues_exactly_the_requested_steps(steps, warmup, use_graph):
    env, hist = StubEnv(), _hist()
    chunk = min(bench.CHUNK, steps)
    r = bench.StepRunner(env, None, hist, None, use_graph=use_graph, chunk=chunk)
    r.prepare(warmup)
    r.prepare(steps, timing=True)
    n_graphs = len(env.captured)
    r.run(warmup)
    assert env._tick == warmup
    for region in range(3):
        segs = r.run(steps)
        assert env._tick == warmup + (region + 1) * steps
        assert bench.episodes_ended(hist, segs, np) > 0           # every segment wrote its rows
        assert sum(s for _, _, s, _ in segs) == steps
    assert r.steps_run == warmup + 3 * steps
    assert len(env.captured) == n_graphs, "no graph may be captured inside a timed region"
    assert all(how == ("graph" if use_graph else "eager") for how, _ in env.calls)
    assert r.launch == ("hipGraph" if use_graph else "eager")
    if use_graph:
        assert max(env.captured) <= chunk


def test_one_block_regions_carry_the_launch_events():
    """run(clock=True): a region of one block is queued by ONE prepared call of plain launches with the events attached to
    its first and last launch (no graph); longer regions, and regions run without the clock, stay on their graphs"""
    env, hist = StubEnv(), _hist()
    r = bench.StepRunner(env, None, hist, None, use_graph=True, chunk=bench.CHUNK, launch_events=object())
    assert r.clocked_by_launch_events(20) and r.clocked_by_launch_events(bench.CHUNK)
    assert not r.clocked_by_launch_events(bench.CHUNK + 1) and not r.clocked_by_launch_events(2000)
    r.prepare(5)
    r.prepare_clocked(20)
    assert env.captured == [5] and env.prepared == [20]
    r.run(5)                                  # warm-up: a graph, as before
    order = []
    r.run(20, before_first_launch=lambda: order.append("e0"), after_last_launch=lambda: order.append("e1"), clock=True)
    assert env.calls == [("graph", 5), ("prepared+events", 20)] and order == ["e0", "e1"] and env._tick == 25
    assert (hist[0][:20, 0] == np.arange(6, 26)).all()          # every step wrote its own done-mask row, in order
    r.prepare(250)
    r.run(250, clock=True)                    # three blocks: graphs, the stream events are the clock
    assert [how for how, _ in env.calls[2:]] == ["graph"] * 3
    plain = bench.StepRunner(StubEnv(), None, _hist(), None, use_graph=True, chunk=bench.CHUNK)
    assert not plain.clocked_by_launch_events(20)          # --region-clock stream


def test_runner_exchange_handoffs():
    """a block is gathered behind the segment that completes it (or the region); the step stream waits for the gather
    that last read a buffer before the first segment that overwrites it"""
    env, hist, ex = StubEnv(), _hist(), StubExchange()
    r = bench.StepRunner(env, None, hist, ex, use_graph=True, chunk=bench.CHUNK)
    r.prepare(1234)
    r.run(1234)          # 500 (buf 0) + 500 (buf 1) + 234 (buf 0)
    assert ex.log == [("wait", 0), ("gather", 0), ("wait", 1), ("gather", 1), ("wait", 0), ("gather", 0)]
    assert ex.finals == [False, False, True]          # only the region's last block is published in stream order
    ex.log.clear()
    closing = []
    r.prepare(20)
    r.run(20, after_last_launch=lambda: closing.append(list(ex.log)))
    assert ex.log == [("wait", 0), ("gather", 0)] and ex.finals[-1] is True
    assert closing == [[("wait", 0)]]                 # the closing event goes in behind the launch, ahead of the exchange


def test_median_pick():
    assert bench.pick_median([5.0]) == 0
    assert bench.pick_median([3.0, 1.0, 2.0]) == 2
    assert bench.pick_median([9.0, 1.0, 5.0, 7.0, 3.0]) == 2
    assert bench.pick_median([4.0, 1.0, 3.0, 2.0]) == 3          # lower of the two middle elements (2.0)


def test_argument_checks():
    assert bench.parse(["--steps", "20", "--warmup", "5"]).steps == 20
    assert bench.parse([]).regions == bench.REGIONS and bench.parse([]).regions_rule.startswith("max(5")
    # five regions at least, and at least one default region's worth of timed steps in all
    assert bench.parse(["--steps", "20", "--warmup", "5"]).regions == 100 and bench.regions_for(1) == bench.MAX_REGIONS
    assert [bench.regions_for(k) for k in (20, 100, 399, 400, 401, 2000, 100000)] == [100, 20, 6, 5, 5, 5, 5]
    explicit = bench.parse(["--steps", "20", "--regions", "7"])
    assert explicit.regions == 7 and explicit.regions_rule == "--regions"
    for bad in (["--steps", "0"], ["--warmup", "-1"], ["--regions", "0"], ["--region-clock", "auto"]):
        with pytest.raises(SystemExit):
            bench.parse(bad)
    assert bench.parse([]).region_clock == "stream"          # the graph between two recorded stream events is the default clock
    assert bench.parse(["--region-clock", "launch"]).region_clock == "launch"


def test_traffic_is_null_without_a_matching_build(monkeypatch):
    """roofline.traffic comes from a committed PMC pass of the SAME build or is null with the reason"""
    args = bench.parse([])
    monkeypatch.setattr(bench, "library_tag", lambda: "0" * 16)
    traffic, src = bench.committed_traffic(262144, args)
    assert traffic is None and isinstance(src, str) and src


# ------------------------------------------------------------------ the N > 1 entry launches itself (round 2: `bench.py --gpus N` sys.exit()ed)
class _Args(object):
    def __init__(self, gpus, ranks_on_one_gpu=False, exchange="auto", exchange_note=None, launch_deadline=330.0):
        self.gpus, self.ranks_on_one_gpu = gpus, ranks_on_one_gpu
        self.exchange, self.exchange_note, self.launch_deadline = exchange, exchange_note, launch_deadline


LINE = '{"metric": "x"}\n'


def test_self_launch_is_needed_only_for_a_bare_multi_gpu_command():
    assert bench.needs_self_launch(2, {})
    assert bench.needs_self_launch(8, {"WORLD_SIZE": "1"})
    assert not bench.needs_self_launch(1, {})
    assert not bench.needs_self_launch(2, {"RANK": "0", "WORLD_SIZE": "2"})     # torch.distributed.run is around it
    assert not bench.needs_self_launch(2, {"WORLD_SIZE": "2"})


def test_launch_command_is_the_drivers_own():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], 29511, python="py", script="bench.py")
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                   "--master-port", "29511", "bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"]


def test_self_launch_refuses_more_ranks_than_gpus(capsys):
    ran = []
    rc = bench.self_launch(_Args(2), ["--gpus", "2"], device_count=1, run=lambda cmd, env, deadline: ran.append(cmd) or (0, LINE, False))
    assert rc != 0 and not ran, "must fail before starting anything"
    assert "--gpus 2" in capsys.readouterr().err
    assert bench.self_launch(_Args(2), ["--gpus", "2"], device_count=0, run=lambda cmd, env, deadline: (0, LINE, False)) != 0


def test_self_launch_starts_a_fresh_child_and_relays_its_line_and_status(capsys):
    seen = {}

    def run(cmd, env, deadline):
        seen["cmd"], seen["env"], seen["deadline"] = cmd, env, deadline
        return 7, LINE, False
    rc = bench.self_launch(_Args(2), ["--gpus", "2", "--steps", "3"], device_count=8, run=run)
    assert rc == 7 and capsys.readouterr().out == LINE          # a line arrived: its ranks' status is the status
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["deadline"] == 330.0
    # the rehearsal layout is allowed on a 1-GPU box
    assert bench.self_launch(_Args(2, ranks_on_one_gpu=True), ["--gpus", "2", "--ranks-on-one-gpu"], device_count=1, run=run) == 7


def test_fallback_chain_and_argument_rewriting():
    assert bench.fallback_chain("auto", False) == ["auto", "rccl", "none"]
    assert bench.fallback_chain("auto", True) == ["auto", "none"]          # ranks on one GPU: no RCCL to fall back to
    for explicit in ("ipc", "rccl", "none"):
        assert bench.fallback_chain(explicit, False) == [explicit]         # what was asked for, or a failure
    assert bench.argv_with(["--gpus", "8", "--exchange", "auto", "--steps", "20"], "rccl", "why") == \
        ["--gpus", "8", "--steps", "20", "--exchange", "rccl", "--exchange-note", "why"]
    assert bench.argv_with(["--exchange=ipc", "--exchange-note=old", "--gpus", "2"], "none", None) == ["--gpus", "2", "--exchange", "none"]


def test_self_launch_falls_back_to_fresh_ranks_until_a_line_arrives(capsys):
    """--exchange auto: ranks that die, or hang past their deadline, are followed by FRESH ranks with --exchange rccl and
    then --exchange none; the first line that arrives is relayed with its ranks' status, and the later attempts carry
    what failed before them in --exchange-note"""
    calls = []

    def run(cmd, env, deadline):
        calls.append((cmd, deadline))
        n = len(calls)
        if n == 1:
            return 3, None, False                 # the watchdog's status, no line
        if n == 2:
            return -9, None, True                 # still running at its deadline: killed
        return 0, LINE, False
    rc = bench.self_launch(_Args(8), ["--gpus", "8", "--steps", "20", "--warmup", "5"], device_count=8, run=run)
    out = capsys.readouterr()
    assert rc == 0 and out.out == LINE and len(calls) == 3
    first, second, third = (c for c, _ in calls)
    assert "--exchange" not in first and first[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert second[second.index("--exchange") + 1] == "rccl" and third[third.index("--exchange") + 1] == "none"
    note2, note3 = second[second.index("--exchange-note") + 1], third[third.index("--exchange-note") + 1]
    assert "--exchange auto: ranks exited with status 3" in note2
    assert note2 in note3 and "--exchange rccl: no line after 150 s" in note3
    assert [d for _, d in calls] == [330.0, 150.0, 150.0]
    assert "starting fresh ranks with --exchange rccl" in out.err and "--exchange none" in out.err
    # every attempt failed: non-zero, nothing on stdout
    calls.clear()
    rc = bench.self_launch(_Args(8), ["--gpus", "8"], device_count=8, run=lambda cmd, env, deadline: (calls.append(cmd) or 1, None, False))
    assert rc != 0 and len(calls) == 3 and capsys.readouterr().out == ""
    # an explicit transport is not second-guessed
    calls.clear()
    rc = bench.self_launch(_Args(8, exchange="ipc"), ["--gpus", "8", "--exchange", "ipc"], device_count=8,
                           run=lambda cmd, env, deadline: (calls.append(cmd) or 3, None, False))
    assert rc == 3 and len(calls) == 1


def test_run_ranks_relays_the_line_and_kills_what_outlives_its_deadline():
    import io
    import sys
    import time
    err = io.StringIO()
    status, line, timed_out = bench.run_ranks([sys.executable, "-c", "print('chatter'); print('{\"a\": 1}'); print('more')"],
                                              dict(__import__("os").environ), 30, err=err)
    assert (status, line, timed_out) == (0, '{"a": 1}\n', False) and "chatter" in err.getvalue() and "more" in err.getvalue()
    t0 = time.time()
    status, line, timed_out = bench.run_ranks([sys.executable, "-c", "import time; time.sleep(600)"],
                                              dict(__import__("os").environ), 1.0, err=err)
    assert timed_out and line is None and status != 0 and time.time() - t0 < 30
    status, line, timed_out = bench.run_ranks([sys.executable, "-c", "import sys; sys.exit(3)"], dict(__import__("os").environ), 30, err=err)
    assert (status, line, timed_out) == (3, None, False)


def test_graph_launches_get_the_marshalled_stream():
    """the runner looks its graph up ahead of the region's first event and hands launch() the stream it marshalled once"""
    env, hist = StubEnv(), _hist(20)
    r = bench.StepRunner(env, None, hist, None, use_graph=True, chunk=20)
    r.prepare(20)
    order = []
    r.run(20, before_first_launch=lambda: order.append("event"), after_last_launch=lambda: order.append("event"))
    assert r.graphs[(0, 0, 20)].streams == [None] and order == ["event", "event"]
    r.stream_handle = "the stream"
    r.run(20)
    assert r.graphs[(0, 0, 20)].streams == [None, "the stream"] and env._tick == 40


def test_warmup_replays_the_timed_graphs_or_uploads_them():
    """region 0 must not be the first launch of its graphs: the warm-up budget is spent on whole timed regions where it is
    large enough, and otherwise the timed graphs are uploaded ahead of time -- exactly `warmup` steps either way"""
    assert bench.warmup_runs(5, 20) == [5] and bench.warmup_runs(0, 20) == [] and bench.warmup_runs(200, 2000) == [200]
    assert bench.warmup_runs(20, 20) == [20] and bench.warmup_runs(100, 30) == [30, 30, 30, 10] and bench.warmup_runs(7, 1) == [1] * 7
    for warmup, steps in ((5, 20), (200, 2000), (100, 30), (0, 1), (250, 230)):
        assert sum(bench.warmup_runs(warmup, steps)) == warmup
    # the driver's flags: the 20-step graph is not replayed by a 5-step warm-up -> uploaded
    env, hist = StubEnv(), _hist(20)
    r = bench.StepRunner(env, None, hist, None, use_graph=True, chunk=20)
    warm = bench.warmup_runs(5, 20)
    for w in warm:
        r.prepare(w)
    r.prepare(20)
    assert r.upload_unplayed(20, warm) == 1 and r.graphs[(0, 0, 20)].uploads == 1 and r.graphs[(0, 0, 5)].uploads == 0
    # ... or replayed once, untimed, and the batch put back where the warm-up left it (--first-replay rollback)
    r.run(5)
    assert env._tick == 5 and r.steps_run == 5
    assert r.replay_unplayed_and_roll_back(20, warm) == 1
    assert env._tick == 5 and r.steps_run == 5 and env.calls[-3:] == [("snapshot", 5), ("graph", 20), ("restore", 5)]
    assert r.replay_unplayed_and_roll_back(5, warm) == 0 and env.calls[-1] == ("restore", 5)      # nothing unplayed: nothing touched
    # defaults: 2 of the 2000-step region's 10 graphs are the warm-up's own
    env, hist = StubEnv(), _hist()
    r = bench.StepRunner(env, None, hist, None, use_graph=True, chunk=bench.CHUNK)
    r.prepare(200)
    r.prepare(2000)
    assert r.upload_unplayed(2000, [200]) == 8
    # warm-up >= steps: the timed graph itself is replayed, nothing to upload
    env, hist = StubEnv(), _hist(30)
    r = bench.StepRunner(env, None, hist, None, use_graph=True, chunk=30)
    warm = bench.warmup_runs(100, 30)
    for w in warm + [30]:
        r.prepare(w)
    assert r.upload_unplayed(30, warm) == 0
    assert bench.StepRunner(StubEnv(), None, _hist(), None, use_graph=False, chunk=100).upload_unplayed(20, [5]) == 0


def test_issue_bound_needs_a_budget_of_the_same_build(monkeypatch):
    args = bench.parse([])
    monkeypatch.setattr(bench, "library_tag", lambda: "0" * 16)
    us, src = bench.issue_bound(262144, args, 5000.0)
    assert us is None and isinstance(src, str) and src


def test_exchange_blocks_counts_the_blocks_of_a_region():
    assert bench.exchange_blocks(20, 20, 20) == 1
    assert bench.exchange_blocks(2000, 100, 500) == 4
    assert bench.exchange_blocks(230, 100, 230) == 1
    assert bench.exchange_blocks(0, 100, 500) == 0
    assert bench.exchange_blocks(501, 100, 500) == 2


def test_popcount_without_numpy2(monkeypatch):
    w = np.array([0, 1, 3, 2 ** 63, 2 ** 64 - 1], dtype=np.uint64)
    assert bench.popcount_words(w, np) == 0 + 1 + 2 + 1 + 64

    class OldNumpy(object):
        def __getattr__(self, name):
            if name == "bitwise_count":
                raise AttributeError(name)
            return getattr(np, name)
    assert bench.popcount_words(w.view(np.int64), OldNumpy()) == 68


# ------------------------------------------------------------------ round 5: one coherent line, and the N > 1 legs
def test_region_outliers_name_the_slowest_region():
    out = bench.region_outliers([5.0, 5.1, 4.9, 13.0, 5.0, 7.4])
    assert out["n"] == 1 and out["slowest"] == {"region": 3, "us_per_step": 13.0, "x_median": 13.0 / 5.05}
    assert bench.region_outliers([5.0] * 8)["n"] == 0 and bench.region_outliers([])["slowest"] is None


def test_every_fraction_sits_beside_the_rate_it_follows_from():
    """achieved = A x steps per second -- for the wall clock (value <-> frac_by_wall) and for the events
    (value_by_events <-> frac), on one GPU and on eight (weak scaling: the fraction is per GPU)"""
    for world in (1, 8):
        r = bench.summarize_regions([1.2e-4, 1.0e-4, 1.1e-4], [0.11, 0.09, 0.10], 20, 262144, world, 62)
        assert r["regions"] == 3 and abs(r["wall_us_per_step"] - 5.5) < 1e-9 and abs(r["event_us_per_step"] - 5.0) < 1e-9
        assert abs(r["frac_by_wall"] - 62 * r["value"] / world / 1e9 / bench.HBM_PEAK_GBPS) < 1e-12
        assert abs(r["frac"] - 62 * r["value_by_events"] / world / 1e9 / bench.HBM_PEAK_GBPS) < 1e-12
        assert abs(r["value_by_events"] - world * 262144 / 5.0e-6) < 1.0


def test_the_untimed_replay_is_dropped_where_region_zero_cannot_move_the_median():
    assert bench.rolls_back_first_replay("rollback", True, False, 5)
    assert not bench.rolls_back_first_replay("rollback", True, False, bench.regions_for(20))     # the driver's command: 100 regions
    assert not bench.rolls_back_first_replay("upload", True, False, 5)
    assert not bench.rolls_back_first_replay("rollback", False, False, 5) and not bench.rolls_back_first_replay("rollback", True, True, 5)
    assert bench.parse(["--steps", "20"]).ab_regions == 100 and bench.parse([]).ab_regions == bench.AB_MIN_REGIONS
    assert bench.parse(["--ab-regions", "3"]).ab_regions == 3


def test_legs_of_the_exchange_ab():
    assert bench.exchange_ab_legs("ipc", True) == ["step_only", "rccl"]          # auto settled on IPC: RCCL is measured too
    assert bench.exchange_ab_legs("rccl", True) == ["step_only", "ipc"]
    assert bench.exchange_ab_legs(None, False) == ["ipc", "rccl"]                # the main regions were the step path alone


def test_region_clock_counts_and_brackets_on_the_cpu():
    env, hist, ex = StubEnv(), _hist(20), StubExchange()
    ex.finish = lambda: ex.log.append(("finish",))
    ex.note_fence = lambda: ex.log.append(("fence",))
    r = bench.StepRunner(env, None, hist, ex, use_graph=True, chunk=20)
    r.prepare(20)
    seen = []
    clock = bench.RegionClock(r, 20, on_region=lambda segs: seen.append(len(segs)))
    walls, events, segs = clock.run(3, ex)
    assert env._tick == 60 and len(walls) == len(events) == 3 and seen == [1, 1, 1] and clock.regions_run == 3
    assert all(abs(e - w * 1e3) < 1e-12 for e, w in zip(events, walls))           # no device: the wall clock is the event clock
    per_region = [("wait", 0), ("gather", 0), ("finish",), ("fence",)]
    assert ex.log == per_region * 3
    ex.log.clear()
    r.exchange = None
    clock.run(2, None)
    assert env._tick == 100 and ex.log == []


class _Guard(object):
    def __init__(self):
        self.stages = []

    def __setattr__(self, name, value):
        if name == "stage":
            self.stages.append(value)
        object.__setattr__(self, name, value)


def test_exchange_ab_bookkeeping_never_touches_what_was_measured():
    env, hist = StubEnv(), _hist(20)
    main = StubExchange()
    r = bench.StepRunner(env, None, hist, main, use_graph=True, chunk=20)
    r.prepare(20)
    clock = bench.RegionClock(r, 20)
    closed, abandoned = [], []

    class Leg(StubExchange):
        def __init__(self, name, fail_in_run=False):
            StubExchange.__init__(self)
            self.name, self.fail_in_run = name, fail_in_run

        def finish(self):
            if self.fail_in_run:
                raise OSError("link down")

        def note_fence(self):
            pass

        def close(self):
            closed.append(self.name)

        def abandon(self):
            abandoned.append(self.name)

    def open_leg(name):
        if name == "ipc":
            raise RuntimeError("cannot map")
        return None if name == "step_only" else Leg(name, fail_in_run=(name == "broken"))

    def agree(error, what):
        if error is not None:
            raise RuntimeError("%s failed on 1 of 1 ranks (%s)" % (what, error))

    guard, out = _Guard(), {"main": "x"}
    got = bench.run_exchange_ab(clock, r, ["step_only", "ipc", "rccl", "broken", "never"], open_leg, 2,
                                lambda w, e: {"n": len(w)}, agree, guard=guard, check=lambda ex, segs: {"checked": ex.name}, out=out)
    assert got is out and out["main"] == "x"
    assert out["step_only"] == {"n": 2}                                           # no exchange: nothing to check or close
    assert out["ipc"] == {"error": "set-up: RuntimeError: cannot map"}            # recorded, skipped, the legs go on
    assert out["rccl"] == {"n": 2, "checked": "rccl"} and closed == ["rccl"]
    assert "link down" in out["broken"]["error"] and out["error"] == "legs stopped in 'broken'" and abandoned == ["broken"]
    assert "never" not in out                                                      # a leg that failed while running ends the legs
    assert r.exchange is main                                                      # ... and the runner is back on the main transport
    assert env._tick == r.steps_run == 2 * 20 * 2 + 20                             # two legs ran 2 regions; the broken one's first region
    assert guard.stages[0] == "exchange_ab leg 'step_only': set-up" and "exchange_ab leg 'rccl': closing its exchange" in guard.stages
