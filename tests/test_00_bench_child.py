"""bench.py end to end on the GPU, as the driver runs it: in a FRESH child process, with the driver's own flags.

This file sorts first on purpose: the children are started (fork + exec) before this pytest process has made its
first HIP call -- a process that has initialised the GPU must not be the one that execs on the GPU boxes.  Nothing
here touches the GPU in-process (torch.cuda.device_count() does not initialise it)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout=900, faults=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("AQUA_TEST_EXCHANGE_FAIL", None)
    if faults:
        env["AQUA_TEST_EXCHANGE_FAIL"] = faults            # honoured by aquaticgymenv_amd/sharded.py only
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, "rc %d\n--- stdout\n%s\n--- stderr\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), \
        "stdout must hold exactly one line, the JSON line (RCCL's banner and everything else belong on stderr):\n%s" % p.stdout[-2000:]
    return json.loads(lines[0])


def _check_line(r, steps, warmup, n_gpus=1, envs=262144, timings_mean_something=True, clock=None):
    clock = clock or "stream"                                      # bench.py's default --region-clock
    assert r["steps"] == steps and r["warmup"] == warmup and r["n_gpus"] == n_gpus
    assert r["unit"] == "env-steps/s" and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["value"] > (1.0e7 if timings_mean_something else 0.0), "below the 10 M env-steps/s target: %r" % r["value"]
    assert abs(r["value"] - n_gpus * envs * steps / (r["ms_per_step"] * 1e-3 * steps)) <= 1e-6 * r["value"]
    assert r["config"]["baseline_config"] == "configs[2]" and str(envs) in r["config"]["workload"]
    assert r["config"]["ranks_seen"] == n_gpus and len(r["config"]["devices"]) == n_gpus
    assert r["config"]["launch"] == ("eager" if clock == "launch" else "hipGraph")
    regions = max(5, -(-2000 // steps))              # bench.regions_for(): five at least, 2 000 timed steps at least
    assert len(r["regions_ms"]) == r["regions"] == regions and r["regions_rule"].startswith("max(5")
    assert sorted(r["regions_ms"])[(regions - 1) // 2] == pytest.approx(r["ms_per_step"] * steps)
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"])
    assert roof["achieved"] == pytest.approx(62 * envs / (roof["launch_us"] * 1e-6) / 1e9)
    assert roof["launch_us_events"] == clock and len(roof["launch_us_regions"]) == regions
    assert len(roof["launch_us_stream_events_regions"]) == regions and 0.0 < roof["frac_by_stream_events"] < 1.0
    first = roof["first_regions"]                      # the first five regions on their own
    assert first["n"] == 5 and first["frac"] == pytest.approx(62 * envs / (first["launch_us"] * 1e-6) / 1e9 / 8000.0)
    assert first["launch_us"] == pytest.approx(sorted(roof["launch_us_regions"][:5])[2])
    if clock == "launch" and timings_mean_something:
        # first kernel start -> last kernel end lies INSIDE the two stream events recorded around the same launches
        assert all(a <= b * 1.001 for a, b in zip(roof["launch_us_regions"], roof["launch_us_stream_events_regions"]))
        assert roof["frac_by_stream_events"] <= roof["frac"] * 1.001
    elif clock == "stream":
        assert roof["launch_us_regions"] == roof["launch_us_stream_events_regions"]
    assert (0.05 if timings_mean_something else 0.0) < roof["frac"] < 1.0
    assert "traffic" in roof and "traffic_source" in roof
    assert r["sanity"]["steps_queued"] == warmup + regions * steps
    assert r["sanity"]["episodes_ended_last_region"] > 0
    # what `value` counts: every world in every step; the restart ticks of the next-step mode are reported beside it
    frac = r["sanity"]["restart_ticks_fraction"]
    assert frac == pytest.approx(r["sanity"]["episodes_ended_last_region"] / float(steps * envs)) and 0.0 < frac < 0.1
    assert r["sanity"]["live_world_steps_per_s"] == pytest.approx(r["value"] * (1.0 - frac))
    # a timed graph the warm-up does not replay: replayed once untimed and taken back only where region 0 could move the
    # median (fewer than 20 regions); the line says how many steps ran beyond the W warm-up steps
    rolled = warmup < steps and regions < 20 and clock != "launch"
    assert r["config"]["timed_graph_first_replay"].startswith(
        "one untimed replay rolled back" if rolled else ("hipGraphUpload" if warmup < steps else "warm-up")) or clock == "launch"
    assert r["config"]["untimed_steps_beyond_warmup"] == (steps if rolled else 0)
    assert "issue_bound_us" in roof and "issue_bound_source" in roof
    # one coherent line (round 5): each rate beside the fraction that follows from it, achieved = A x steps per second
    assert roof["frac"] == pytest.approx(62 * r["value_by_events"] / n_gpus / 1e9 / 8000.0)
    assert roof["frac_by_wall"] == pytest.approx(62 * r["value"] / n_gpus / 1e9 / 8000.0)
    assert roof["frac_live"] == pytest.approx(roof["frac"] * (1.0 - frac))
    assert r["sanity"]["live_world_steps_per_s_by_events"] == pytest.approx(r["value_by_events"] * (1.0 - frac))
    for which in ("by_events", "by_wall"):
        o = r["regions_outliers"][which]
        assert 0 <= o["n"] < regions and o["slowest"]["x_median"] >= 1.0 and 0 <= o["slowest"]["region"] < regions
    ko = roof["kernel_only"]
    assert set(ko) >= {"avg_ns", "frac", "source"} and (ko["avg_ns"] is None or 0.0 < ko["frac"] < 1.0)


@pytest.mark.gpu
def test_bench_with_the_drivers_flags_in_a_child_process():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (round 1's driver command, which asserted) prints the line"""
    r = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "1"])
    _check_line(r, 20, 5)
    for key in ("cpu_baseline", "cpu_baseline_1core", "cpu_baseline_c"):
        b = r[key]
        assert b["value"] > 0 and b["unit"] == "env-steps/s" and b["kind"] == "port" and b["cores"] >= 1 and b["sample"]
    assert r["cpu_baseline"]["cores"] >= r["cpu_baseline_1core"]["cores"] == 1


@pytest.mark.gpu
def test_bench_odd_step_counts_in_a_child_process():
    """one step without warm-up, and a count that needs full chunks + a remainder"""
    r = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r["steps"] == 1 and r["regions"] == 400 and r["sanity"]["steps_queued"] == 400      # (bench.MAX_REGIONS one-step regions)
    r = _run([sys.executable, "bench.py", "--steps", "230", "--warmup", "7", "--no-cpu-baseline"])
    _check_line(r, 230, 7)


@pytest.mark.gpu
def test_bench_diagnostic_flags_in_a_child_process():
    """the flags the round's measurements were taken with: event-record nodes beside the stream events, the copy engines for
    the done-mask copies, a pause between regions, the eager launch loop"""
    r = _run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--force-exchange",
              "--copy-engine", "dma", "--graph-node-events", "--settle-us", "100"])
    _check_line(r, 20, 5)
    nodes = r["roofline"]["launch_us_graph_nodes_regions"]
    assert len(nodes) == 100 and all(v and v > 1.0 for v in nodes)
    assert r["config"]["done_mask_exchange_kind"] == "ipc" and r["config"]["done_mask_copy_engine"] == "dma"
    assert r["sanity"]["done_mask_exchange_last_block"]["own_block_intact"] is True
    r = _run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--region-clock", "launch"])
    _check_line(r, 20, 5, clock="launch")               # the events attached to the region's first and last launch
    r = _run([sys.executable, "bench.py", "--steps", "30", "--warmup", "3", "--no-cpu-baseline", "--eager"])
    assert r["config"]["launch"] == "eager" and r["steps"] == 30 and r["sanity"]["steps_queued"] == 3 + 67 * 30


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_executes_the_rccl_branch():
    """one rank under torch.distributed.run: init_process_group("nccl"), the barriers, the MAX all-reduce of the region
    times and the done-mask all_gather_into_tensor (RCCL, side stream, double buffer) all execute on the device.
    No 1 -> 8 GPU curve is measured here: the driver does that on an 8-GPU node."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "120", "--warmup", "5",
              "--no-cpu-baseline", "--force-exchange"])
    _check_line(r, 120, 5)
    assert r["config"]["done_mask_exchange"] is True and r["config"]["done_mask_exchange_kind"] == "ipc"
    assert r["config"]["process_group"].startswith("nccl")
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5",
              "--no-cpu-baseline", "--force-exchange", "--exchange", "rccl"])
    _check_line(r, 20, 5)
    assert r["config"]["done_mask_exchange_kind"] == "rccl"


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it (the form round 2's bench.py sys.exit()ed on): it starts
    torch.distributed.run itself, as a fresh child, and relays rank 0's line.  On this 1-GPU box the two ranks share cuda:0
    (--ranks-on-one-gpu: gloo rendezvous, the done mask by IPC peer copies between the two processes) -- the N > 1 path end
    to end on device tensors with world size 2; the timings of two processes on one GPU mean nothing and are not checked."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--ranks-on-one-gpu", "--envs", "65536", "--steps", "20", "--warmup", "5",
              "--no-cpu-baseline"])
    _check_line(r, 20, 5, n_gpus=2, envs=65536, timings_mean_something=False)
    cfg = r["config"]
    assert cfg["done_mask_exchange"] is True and cfg["done_mask_exchange_kind"] == "ipc"
    assert cfg["global_worlds"] == 2 * 65536 and cfg["parallelism"] == "range-partition x2"
    check = r["sanity"]["done_mask_exchange_last_block"]
    assert check["own_block_intact"] is True and len(check["episodes_in_peer_blocks"]) == 1
    assert check["episodes_in_peer_blocks"][0] > 0, "rank 1's done masks must have arrived at rank 0"
    assert "cpu_baseline" not in r


@pytest.mark.gpu
def test_two_ranks_time_the_step_path_alone_and_the_other_transport_behind_the_main_regions():
    """round 5: ONE run with N > 1 measures more than the transport `auto` settled on -- extras.exchange_ab carries the main
    transport's figures (the line's own), a `step_only` leg (no exchange at all: what the kernels scale like) and a leg for
    the OTHER transport (here RCCL, which cannot exist with two ranks on one device: recorded as an error, the line
    unharmed); per_rank shows every rank's own event clock"""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--ranks-on-one-gpu", "--envs", "65536", "--steps", "20", "--warmup", "5",
              "--no-cpu-baseline", "--ab-regions", "25"])
    _check_line(r, 20, 5, n_gpus=2, envs=65536, timings_mean_something=False)
    ab = r["extras"]["exchange_ab"]
    assert ab["main"] == "ipc" and ab["regions_per_leg"] == 25 and "error" not in ab
    assert ab["ipc"]["regions"] == 100 and ab["ipc"]["own_block_intact"] is True
    assert ab["ipc"]["value"] == pytest.approx(r["value"]) and ab["ipc"]["frac"] == pytest.approx(r["roofline"]["frac"])
    so = ab["step_only"]
    assert so["regions"] == 25 and so["value"] > 0 and so["event_us_per_step"] > 0 and "own_block_intact" not in so
    assert "RCCL refuses two ranks on one device" in ab["rccl"]["error"] and ab["rccl"]["error"].startswith("set-up:")
    assert [p["rank"] for p in r["per_rank"]] == [0, 1] and all(p["event_us_per_step"] > 0 and "cuda:0" in p["device"] for p in r["per_rank"])
    assert r["sanity"]["steps_queued"] == 5 + 100 * 20          # the line's own regions; the legs ran behind them


@pytest.mark.gpu
def test_one_rank_under_the_launcher_measures_rccl_behind_an_ipc_main_and_the_reverse():
    """the RCCL leg for real (one rank under torch.distributed.run, nccl process group): main regions on IPC peer copies,
    then the step path alone, then RCCL's all-gather -- and the other way round when RCCL is asked for"""
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
            "--force-exchange", "--ab-regions", "30"]
    r = _run(base)
    _check_line(r, 20, 5)
    ab = r["extras"]["exchange_ab"]
    assert ab["main"] == "ipc" and "error" not in ab
    for leg in ("step_only", "rccl"):
        assert ab[leg]["regions"] == 30 and 0.05 < ab[leg]["frac"] < 1.0, (leg, ab[leg])
    assert ab["rccl"]["own_block_intact"] is True and ab["rccl"]["episodes_in_peer_blocks"] == []
    assert len(r["per_rank"]) == 1 and r["per_rank"][0]["event_us_per_step"] == pytest.approx(r["roofline"]["launch_us"])
    base[base.index("--master-port") + 1] = str(_free_port())
    r = _run(base + ["--exchange", "rccl"])
    ab = r["extras"]["exchange_ab"]
    assert ab["main"] == "rccl" and ab["ipc"]["own_block_intact"] is True and ab["step_only"]["regions"] == 30


@pytest.mark.gpu
def test_an_extra_leg_that_stalls_cannot_cost_the_line():
    """the set-up of an extra leg never returns on the highest rank (injected): past --ab-deadline rank 0 prints the line
    with the main regions untouched and extras.exchange_ab.error naming the stage; every rank leaves with status 0"""
    import time
    t0 = time.time()
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--ranks-on-one-gpu", "--envs", "65536", "--steps", "20", "--warmup", "5",
              "--no-cpu-baseline", "--ab-deadline", "12"], faults="ab-stall")
    assert time.time() - t0 < 240
    _check_line(r, 20, 5, n_gpus=2, envs=65536, timings_mean_something=False)
    ab = r["extras"]["exchange_ab"]
    # (rank 0 itself had gone on into the leg's first region and sat in its barrier)
    assert "deadline of 12 s passed in stage: exchange_ab leg 'step_only'" in ab["error"]
    assert ab["ipc"]["own_block_intact"] is True and "step_only" not in ab


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus_without_hanging():
    import torch
    if torch.cuda.device_count() >= 2:      # (device_count() does not initialise HIP)
        pytest.skip("a multi-GPU box runs this command for real")
    env = dict(os.environ)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_ipc_done_mask_exchange_between_two_processes_on_one_gpu():
    """DoneMaskExchange(kind="ipc") for real: three processes on cuda:0 map each other's receive buffers
    (hipIpcGetMemHandle / hipIpcOpenMemHandle through the C ABI) and publish blocks by device-to-device copies"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join("tests", "_ipc_child.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, "rc %d\n--- stdout\n%s\n--- stderr\n%s" % (p.returncode, p.stdout[-3000:], p.stderr[-3000:])
    assert "ipc exchange ok: 3 ranks" in p.stdout


# ------------------------------------------------------------------ first contact cannot end without a line (round 4)
_TWO_ON_ONE = [sys.executable, "bench.py", "--gpus", "2", "--ranks-on-one-gpu", "--envs", "65536", "--steps", "20", "--warmup", "5",
               "--no-cpu-baseline"]


@pytest.mark.gpu
@pytest.mark.parametrize("faults,soft,want", [
    ("open", "45", "injected fault: hipIpcOpenMemHandle failed"),
    ("probe", "45", "did not arrive intact"),
    ("stall", "8", "still in stage 'injected fault: a set-up call that never returns' after 8 s"),
])
def test_a_transport_that_fails_or_stalls_does_not_take_the_line_with_it(faults, soft, want):
    """two ranks on the one GPU, the IPC exchange made to fail at mapping, at the probe, or to STALL (a call that never
    returns) on rank 1: both ranks drop it together inside their processes (no RCCL with two ranks on one device: the run
    goes on without an exchange), the line arrives, and it says what happened"""
    r = _run(_TWO_ON_ONE + ["--soft-deadline", soft], faults=faults)
    _check_line(r, 20, 5, n_gpus=2, envs=65536, timings_mean_something=False)
    cfg = r["config"]
    assert cfg["done_mask_exchange"] is False and cfg["done_mask_exchange_kind"] is None
    assert want in cfg["done_mask_exchange_note"] and "NO done-mask exchange in this run" in cfg["done_mask_exchange_note"]
    assert r["sanity"]["done_mask_exchange_last_block"] is None


@pytest.mark.gpu
def test_ranks_that_hang_are_ended_and_replaced_by_fresh_ranks_without_the_exchange():
    """the set-up of the FIRST set of ranks hangs where no soft deadline reaches (injected): their watchdog ends them with
    status 3 inside --setup-deadline, bench.py's self-launch starts fresh ranks with --exchange none, and their line carries
    the story"""
    import time
    t0 = time.time()
    r = _run(_TWO_ON_ONE + ["--setup-deadline", "10", "--rendezvous-timeout", "20"], faults="hard-stall")
    assert time.time() - t0 < 300
    _check_line(r, 20, 5, n_gpus=2, envs=65536, timings_mean_something=False)
    cfg = r["config"]
    assert cfg["done_mask_exchange"] is False
    assert "--exchange auto: ranks exited with status" in cfg["done_mask_exchange_note"]


@pytest.mark.gpu
def test_one_rank_under_the_launcher_falls_back_to_rccl_inside_the_process():
    """under torch.distributed.run (what the driver uses for N > 1) nobody starts fresh ranks: an IPC transport that
    fails is replaced by RCCL's all-gather inside the process, and by nothing when that fails too"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
           "--force-exchange"]
    r = _run(cmd, faults="open")
    _check_line(r, 20, 5)
    assert r["config"]["done_mask_exchange_kind"] == "rccl" and "injected fault" in r["config"]["done_mask_exchange_note"]
    assert r["sanity"]["done_mask_exchange_last_block"]["own_block_intact"] is True
    cmd[cmd.index("--master-port") + 1] = str(_free_port())
    r = _run(cmd, faults="open,rccl")
    _check_line(r, 20, 5)
    assert r["config"]["done_mask_exchange"] is False and "rccl unavailable" in r["config"]["done_mask_exchange_note"]
