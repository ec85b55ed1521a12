"""Multi-process (world_size 2, gloo, CPU) tests of the N > 1 path: range partition + the done-mask
all-gather.  The worlds themselves are stepped by the CPU oracle here (the HIP kernels need a GPU);
what is under test is the host logic that bench.py and a multi-GPU user run around the kernels:
shard_range(), DoneMaskExchange and the shard-invariance of the Philox keying (global world index)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aquaticgymenv_amd.sharded import DoneMaskExchange, Watchdog, injected_faults, open_exchange, shard_range, unpack_done_words


def test_shard_range_partitions_exactly():
    for total in (1, 63, 64, 65, 1000, 4096, 262144, 2097152, 2097153):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            covered = 0
            for o, c in spans:                       # non-empty shards tile [0, total) in rank order
                if c:
                    assert o == covered
                    covered += c
            assert covered == total
            assert all(o % 64 == 0 for o, _ in spans)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pack(done):
    n = done.shape[0]
    pad = (-n) % 64
    bits = np.concatenate([done.astype(np.uint8), np.zeros(pad, dtype=np.uint8)])
    return np.packbits(bits, bitorder="little").view(np.uint64).astype(np.uint64)


def _worker(rank, world, port, total, steps, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.aqua_oracle import COracle
        from aquaticgymenv_amd import presets
        orc = COracle()
        off, cnt = shard_range(total, world, rank)
        words = (shard_range(total, world, 0)[1] + 63) // 64          # the largest shard: every rank pads to it
        st = np.zeros((7, cnt), dtype=np.float32)
        tt = np.zeros(cnt, dtype=np.int32)
        orc.reset(st, tt, obstacles=presets.BENCH8, seed=11, tick=0, env_offset=off)
        ex = DoneMaskExchange(steps, words, "cpu")
        local = torch.zeros((steps, words), dtype=torch.int64)
        for t in range(steps):
            _, _, term, _ = orc.rollout_f32(st, tt, 1, obstacles=presets.BENCH8, seed=11, tick0=1 + t, env_offset=off)
            w = _pack(term != 0)
            local[t, : w.shape[0]] = torch.from_numpy(w.view(np.int64))
        slot = ex.gather_async(local)
        ex.wait(slot)
        ex.finish()
        g = ex.gathered[slot].numpy()
        assert g.shape == (world, steps, words)
        np.save(os.path.join(tmpdir, "gathered_%d.npy" % rank), g)
        np.save(os.path.join(tmpdir, "state_%d.npy" % rank), st)
        # every rank holds the same gathered block
        t0 = torch.from_numpy(g.copy())
        dist.broadcast(t0, src=0)
        assert np.array_equal(t0.numpy(), g)
    finally:
        dist.destroy_process_group()


def test_two_rank_rollout_matches_single_process(tmp_path):
    total, steps, world = 5000, 12, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle.aqua_oracle import COracle
    from aquaticgymenv_amd import presets
    orc = COracle()
    st = np.zeros((7, total), dtype=np.float32)
    tt = np.zeros(total, dtype=np.int32)
    orc.reset(st, tt, obstacles=presets.BENCH8, seed=11, tick=0)
    dones = []
    for t in range(steps):
        _, _, term, _ = orc.rollout_f32(st, tt, 1, obstacles=presets.BENCH8, seed=11, tick0=1 + t)
        dones.append((term != 0).astype(np.uint8))
    g = np.load(tmp_path / "gathered_0.npy")
    assert np.array_equal(g, np.load(tmp_path / "gathered_1.npy"))
    spans = [shard_range(total, world, r) for r in range(world)]
    for t in range(steps):
        got = np.concatenate([unpack_done_words(g[r, t], spans[r][1]) for r in range(world)])
        assert np.array_equal(got, dones[t])
    # shard invariance of the state itself (Philox keyed by the global world index)
    whole = np.concatenate([np.load(tmp_path / ("state_%d.npy" % r)) for r in range(world)], axis=1)
    assert np.array_equal(whole, st)


class _MaskEnv(object):
    """what bench.StepRunner needs of an environment, on the CPU: step t of rank r writes the word r * 1000 + t + 1 into
    its done-mask row (so a gathered block says who wrote which row when)"""

    def __init__(self, rank):
        self.rank, self._tick = rank, 0

    def rollout(self, steps, actions=None, keep_all=False, done_history=None):
        for i in range(steps):
            done_history[i, :] = self.rank * 1000 + self._tick + i + 1
        self._tick += steps


def _bench_worker(rank, world, port, steps, warmup):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        words = 8
        chunk = min(bench.CHUNK, steps)
        block_rows = max(chunk, min(bench.GATHER_EVERY * bench.CHUNK, steps))
        hist = [torch.zeros((block_rows, words), dtype=torch.int64) for _ in range(2)]
        ex = DoneMaskExchange(block_rows, words, "cpu")
        assert ex.collective and ex.world == world
        env = _MaskEnv(rank)
        runner = bench.StepRunner(env, None, hist, ex, use_graph=False, chunk=chunk)
        runner.run(warmup)
        ex.finish()
        for region in range(3):
            tick0 = env._tick
            segs = runner.run(steps)
            ex.finish()
            assert env._tick == tick0 + steps and sum(s for _, _, s, _ in segs) == steps
            # the block gathered last holds, for EVERY rank, the rows of the region's last block
            buf, row0, s, gathered = segs[-1]
            assert gathered
            block = ex.gathered[ex.last_slot()]
            first_row_tick = tick0 + steps - (row0 + s)          # tick of row 0 of that block
            for r in range(world):
                want = r * 1000 + first_row_tick + torch.arange(1, row0 + s + 1, dtype=torch.int64)
                assert torch.equal(block[r, :row0 + s, 0], want), (rank, region, r)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("steps,warmup", [(20, 5), (230, 7), (1234, 0)])
def test_bench_runner_exchanges_done_masks_between_two_ranks(steps, warmup):
    """bench.py's step queue with the real DoneMaskExchange over gloo, two ranks: the driver's short regions (one block of
    20 rows), a region of full chunks + remainder, and one that wraps the double buffer"""
    mp.spawn(_bench_worker, args=(2, _free_port(), steps, warmup), nprocs=2, join=True)


def test_exchange_world_size_one_cpu():
    ex = DoneMaskExchange(3, 4, "cpu")
    x = torch.arange(12, dtype=torch.int64).reshape(3, 4)
    slot = ex.gather_async(x)
    ex.finish()
    assert torch.equal(ex.gathered[slot][0], x)
    with pytest.raises(ValueError):
        ex.gather_async(torch.zeros((2, 4), dtype=torch.int64))


def test_ipc_block_offsets_tile_the_receive_buffer():
    """kind="ipc": rank s writes its [steps][words] block of slot k at ((k * world + s) * steps * words) * 8 bytes of every
    rank's [slots][world][steps][words] buffer: the blocks of all (slot, sender) pairs tile it without overlap"""
    from aquaticgymenv_amd.sharded import ipc_block_offset
    world, steps, words, slots = 3, 5, 4, 2
    size = steps * words * 8
    offs = sorted(ipc_block_offset(k, s, world, steps, words) for k in range(slots) for s in range(world))
    assert offs == [i * size for i in range(slots * world)]
    import pytest
    with pytest.raises(ValueError):
        ipc_block_offset(0, 3, world, steps, words)


def test_ipc_kind_has_no_cpu_path():
    import pytest
    from aquaticgymenv_amd.sharded import DoneMaskExchange
    with pytest.raises(RuntimeError):
        DoneMaskExchange(4, 2, "cpu", kind="ipc")
    with pytest.raises(ValueError):
        DoneMaskExchange(4, 2, "cpu", kind="smoke signals")


def _agree_worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = DoneMaskExchange(2, 4, "cpu")
        ex._agree(None, "nothing")                                   # nobody failed: nobody raises
        outcome = "no error"
        try:
            ex._agree(ValueError("cannot map rank 0") if rank == 1 else None, "the mapping")
        except RuntimeError as exc:
            outcome = str(exc)
        with open(os.path.join(tmpdir, "agree%d.txt" % rank), "w") as f:
            f.write(outcome)
        dist.barrier()                                               # both are still in step with each other
    finally:
        dist.destroy_process_group()


def test_a_failure_on_one_rank_is_raised_on_every_rank(tmp_path):
    """the IPC exchange decides 'mapped buffers or RCCL' for the whole job: a rank whose mapping (or probe) failed makes
    EVERY rank raise -- a rank that fell back alone would sit in a collective the others never reach"""
    mp.spawn(_agree_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        text = (tmp_path / ("agree%d.txt" % rank)).read_text()
        assert "the mapping failed on 1 of 2 ranks" in text and "rank 1: ValueError: cannot map rank 0" in text


# ------------------------------------------------------------------ bounded time and fallbacks of the N > 1 set-up (round 4)
def test_watchdog_names_the_stage_and_ends_the_process_unless_cancelled(capsys):
    import time
    fired = []
    w = Watchdog(0.2, "mapping the peers' buffers", on_expire=fired.append)
    w.stage = "probing"                                   # the stage can be moved along while the clock runs
    time.sleep(0.6)
    assert fired == [Watchdog.EXIT_STATUS] and Watchdog.EXIT_STATUS == 3
    err = capsys.readouterr().err
    assert "deadline of 0.2 s passed in stage 'probing'" in err and err.count("\n") == 1      # ONE line
    fired.clear()
    with Watchdog(0.2, "quick", on_expire=fired.append):
        pass
    time.sleep(0.5)
    assert fired == []


def test_fault_injection_is_off_unless_asked_for():
    assert injected_faults({}) == set()
    assert injected_faults({"AQUA_TEST_EXCHANGE_FAIL": "open, rccl"}) == {"open", "rccl"}
    assert open_exchange("none", 4, 2, "cpu") == (None, None, None)
    with pytest.raises(ValueError):
        open_exchange("carrier pigeon", 4, 2, "cpu")


def _fallback_worker(rank, world, port, tmpdir, faults, soft):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    if faults:
        os.environ["AQUA_TEST_EXCHANGE_FAIL"] = faults
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        t0 = time.time()
        ex, kind, note = open_exchange("auto", 3, 4, "cpu", slots=2, soft_deadline_s=soft)
        took = time.time() - t0
        outcome = "%s|%s|%.1f" % (kind, note, took)
        if ex is not None:                    # the transport that was chosen works, on every rank, in step
            local = torch.full((3, 4), rank + 1, dtype=torch.int64)
            slot = ex.gather_async(local)
            ex.fence()
            assert [int(ex.gathered[slot][r, 0, 0]) for r in range(world)] == [1, 2]
        dist.barrier()                        # the default group is untouched by whatever happened in the sandbox
        if "slow" in faults:
            # the set-up thread of the highest rank comes back AFTER the attempt was given up: it must stop at its next stage
            # boundary (no buffer mapped, no collective, no copy beside the transport in use) and say how far it had come
            from aquaticgymenv_amd import sharded
            for th in sharded.ABANDONED_SETUP_THREADS:
                th.join(timeout=20.0)
            alive = any(th.is_alive() for th in sharded.ABANDONED_SETUP_THREADS)
            outcome += "|late=%s|alive=%s" % (";".join(sharded.LATE_SETUP_NOTES), alive)
            if ex is not None:                # ... and the transport in use still works afterwards
                slot = ex.gather_async(torch.full((3, 4), 7 * (rank + 1), dtype=torch.int64))
                ex.fence()
                assert [int(ex.gathered[slot][r, 0, 0]) for r in range(world)] == [7, 14]
        with open(os.path.join(tmpdir, "out%d.txt" % rank), "w") as f:
            f.write(outcome)
    finally:
        from aquaticgymenv_amd import sharded
        if any(th.is_alive() for th in sharded.ABANDONED_SETUP_THREADS):
            os._exit(0)                       # (what bench.py does: a thread of this process sits in a call that never returns)
        dist.destroy_process_group()


@pytest.mark.parametrize("faults,soft,want_kind,want_note", [
    ("", 20.0, "rccl", "ipc unavailable"),                            # no HIP device here: the IPC transport fails on every rank
    ("rccl", 20.0, "None", "NO done-mask exchange in this run"),      # ... and so does the next one: the run goes on without
    ("stall", 3.0, "rccl", "still in stage 'injected fault: a set-up call that never returns' after 3 s"),
])
def test_open_exchange_falls_back_together_on_every_rank(tmp_path, faults, soft, want_kind, want_note):
    """two ranks (gloo, CPU): the IPC set-up fails (no GPU) or STALLS on one rank (injected) inside its sandbox; every rank
    drops it together -- through the untouched default group -- and goes on with the collective transport (gloo stands in
    for RCCL on CPU tensors), or with no exchange at all when that fails too"""
    mp.spawn(_fallback_worker, args=(2, _free_port(), str(tmp_path), faults, soft), nprocs=2, join=True)
    for rank in range(2):
        kind, note, took = (tmp_path / ("out%d.txt" % rank)).read_text().split("|")
        assert kind == want_kind and want_note in note, (rank, kind, note)
        assert float(took) < soft + 15.0
        if faults == "stall":
            assert "rank 1: TimeoutError" in note                      # who stalled, where, is on the record


def test_a_set_up_thread_that_comes_back_late_stops_at_its_next_stage(tmp_path, monkeypatch):
    """open_exchange() used to leave a slow (not stuck) set-up thread running after the soft deadline -- it would
    have gone on to map buffers, run collectives and copy on side streams while the run had moved to the next transport.
    Now the main thread sets a cancel flag once the ranks have agreed to move on, and the late thread stops at its next
    stage boundary.  Two ranks, the highest one's set-up sleeps 4 s past a 1.5-s soft deadline."""
    monkeypatch.setenv("AQUA_TEST_SLOW_S", "4")
    mp.spawn(_fallback_worker, args=(2, _free_port(), str(tmp_path), "slow", 1.5), nprocs=2, join=True)
    for rank in range(2):
        fields = (tmp_path / ("out%d.txt" % rank)).read_text().split("|")
        kind, note = fields[0], fields[1]
        assert kind == "rccl" and "rank 1: TimeoutError: still in stage 'injected fault: a set-up call that returns late'" in note
        late, alive = fields[3], fields[4]
        assert alive == "alive=False"
        if rank == 1:
            assert "cancelled before stage 'starting the IPC set-up'" in late, late
        else:
            assert late == "late="


# ------------------------------------------------------------------ round 5: the extra legs behind the main regions
def _ab_worker(rank, world, port, tmpdir, fail_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        import bench
        from aquaticgymenv_amd.sharded import agree
        steps, words = 20, 8
        hist = [torch.zeros((steps, words), dtype=torch.int64) for _ in range(2)]
        main, kind, _ = open_exchange("rccl", steps, words, "cpu", slots=2)          # (gloo stands in for RCCL on CPU tensors)
        env = _MaskEnv(rank)
        runner = bench.StepRunner(env, None, hist, main, use_graph=False, chunk=steps)
        clock = bench.RegionClock(runner, steps, dist=dist)
        clock.rendezvous(main)
        walls, events, segs = clock.run(3, main)                                      # the "main regions"

        def max_over_ranks(values):
            t = torch.tensor(values, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return [float(v) for v in t]

        class Flaky(object):
            """the second RCCL leg's exchange: what arrived turns out damaged on ONE rank (found by the leg's check, when
            every rank is out of its regions -- a failure INSIDE a region leaves the other ranks in that region's barrier,
            which only a deadline ends: tests/test_00_bench_child.py, --ab-deadline)"""
            def __init__(self, ex):
                self.ex = ex

            def __getattr__(self, name):
                return getattr(self.ex, name)

            def last_slot(self):
                if rank == fail_rank:
                    raise OSError("injected: the last block arrived damaged")
                return self.ex.last_slot()

        def open_leg(name):
            if name == "step_only":
                return None
            ex, _, _ = open_exchange(name.split("#")[0], steps, words, "cpu", slots=2, soft_deadline_s=10.0)
            return Flaky(ex) if name.endswith("#flaky") else ex

        def check(ex, leg_segs):
            got = ex.gathered[ex.last_slot()]
            return {"own_block_intact": bool(torch.equal(got[rank], hist[leg_segs[-1][0]])),
                    "peer_first_words": [int(got[r][0, 0]) for r in range(world) if r != rank]}

        out = {}
        bench.run_exchange_ab(clock, runner, ["step_only", "ipc", "rccl", "rccl#flaky", "rccl"], open_leg, 4,
                              lambda w, e: bench.summarize_regions(max_over_ranks(w), e, steps, 512, world, 62),
                              lambda err, what: agree(dist, None, world, err, what), check=check, out=out)
        assert runner.exchange is main
        dist.barrier()                                        # both ranks are still in step with each other ...
        tick0 = env._tick
        clock.run(1, main)                                    # ... and the main transport still works
        block = main.gathered[main.last_slot()]
        assert [int(block[r, 0, 0]) for r in range(world)] == [r * 1000 + tick0 + 1 for r in range(world)]
        with open(os.path.join(tmpdir, "ab%d.json" % rank), "w") as f:
            json.dump({"out": out, "tick": env._tick, "steps_run": runner.steps_run}, f)
        main.close()
    finally:
        dist.destroy_process_group()


def test_extra_legs_run_in_step_on_two_ranks_and_a_failing_leg_ends_them_together(tmp_path):
    """bench.py's exchange_ab over gloo, two ranks: the step path alone; a transport that cannot be set up (IPC without a
    HIP device: an error on every rank, skipped); the collective transport with its own block check; a leg that FAILS on
    rank 1 inside a timed region -- both ranks stop the legs together (nobody sits in a collective the other left), what
    was measured stays, and the main transport goes on working"""
    import json
    mp.spawn(_ab_worker, args=(2, _free_port(), str(tmp_path), 1), nprocs=2, join=True)
    for rank in range(2):
        rec = json.loads((tmp_path / ("ab%d.json" % rank)).read_text())
        out = rec["out"]
        assert out["step_only"]["regions"] == 4 and "own_block_intact" not in out["step_only"]
        assert out["ipc"]["error"].startswith("set-up:") and "ipc" in out["ipc"]["error"]
        assert out["rccl"]["regions"] == 4 and out["rccl"]["own_block_intact"] is True
        assert out["rccl"]["peer_first_words"][0] % 1000 > 0                       # the peer's rows arrived
        assert "timed regions" in out["rccl#flaky"]["error"] and "rank 1: OSError: injected" in out["rccl#flaky"]["error"]
        assert out["error"] == "legs stopped in 'rccl#flaky'"
        # 3 main regions + 4 (step_only) + 4 (rccl) + the flaky leg's regions up to its failure + 1 closing region
        assert rec["tick"] == rec["steps_run"] and rec["tick"] >= 20 * (3 + 4 + 4 + 1 + 1)
    a = json.loads((tmp_path / "ab0.json").read_text())["out"]
    b = json.loads((tmp_path / "ab1.json").read_text())["out"]
    assert a["rccl"]["value"] == b["rccl"]["value"]                                # the MAX over the ranks is everybody's
