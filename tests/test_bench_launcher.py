"""CPU tests of bench.py's launcher logic (no GPU, no ranks started on a GPU): `--gpus N` without a launcher starts the driver's
own launch line as a CHILD process and relays rank 0's line; `--gpus` that disagrees with WORLD_SIZE is an error, never a silent
n_gpus: 1; the PMC entries bench.py accepts are tied to the kernel sources' contents."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module("bench")


def test_launcher_command_is_the_drivers_launch_line():
    b = _bench()
    cmd, env = b.launcher_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], 29517, python="python3", script="/x/bench.py")
    assert cmd == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                   "--master-port", "29517", "/x/bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        assert k not in env                      # the child's launcher sets them; stale values of the parent must not leak
    assert env["TPNET_BENCH_LAUNCHED_BY"] == str(os.getpid())


def test_pick_result_line_takes_rank_zeros_json():
    b = _bench()
    out = "\n".join(["W0101 torch.distributed.run noise", '{"not": "it"}', "smoke",
                     json.dumps({"metric": b.METRIC, "value": 1.0, "n_gpus": 2, "ranks_seen": 2}), "trailing"])
    line = b.pick_result_line(out)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert b.pick_result_line("nothing here") is None


def _run(args, env_extra, timeout=180):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_flag_that_disagrees_with_world_size_is_an_error():
    r = _run(["--gpus", "3", "--steps", "2", "--warmup", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and "WORLD_SIZE=2" in line["error"] and line["n_gpus"] == 2


def test_gpus_n_without_a_launcher_never_reports_one_gpu():
    """No GPU in this tier: the parent must say so (exit code != 0, an error line with n_gpus = N) -- not run one rank and print
    n_gpus: 1 as round 4's bench.py did."""
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("needs a box without GPUs (the GPU tier runs the real launch: tests/test_sharded.py)")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {})
    assert r.returncode != 0
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] is None


def test_csrc_fingerprint_names_the_kernel_sources():
    b = _bench()
    f1 = b.csrc_fingerprint()
    assert len(f1) == 16 and int(f1, 16) >= 0 and f1 == b.csrc_fingerprint()


def test_workspace_of_a_very_long_stream_is_bounded(hip_lib):
    """the side-by-side plans of a multi-chunk stream are granted only within TPNET_ARENA_MAX_RATIO (3) times
    one chunk's workspace -- a 100 M-edge C2 stream asked for 108 GB, a 64-chunk one for more than the GPU holds."""
    N, d, L, B = 9228, 128, 3, 1000
    row = 2 * L * d * 4
    one = hip_lib.tpnet_stream_workspace_bytes(N, d, L, 256 * 24 * B, B)        # the longest one-chunk stream (256 windows of 24)
    big = hip_lib.tpnet_stream_workspace_bytes(N, d, L, 100_000_000, B)
    assert big <= 3 * one and big < 64 << 30
    for E in (5_000_000, 20_000_000, 64 * 256 * 24 * B):
        assert hip_lib.tpnet_stream_workspace_bytes(N, d, L, E, B) <= 3 * one
    # (a stream of a few chunks still gets the replayable layout: the capped case of test_abi)
    four = hip_lib.tpnet_stream_workspace_bytes_capped(N, d, L, 480 * B, B, 120 * B * row)
    assert four > hip_lib.tpnet_stream_workspace_bytes(N, d, L, 120 * B, B)


def test_stream_schedule_query(hip_lib):
    """tpnet_stream_schedule: what bench.py asks instead of restating the C side's rule (16 batches of <= 2048 edges)."""
    N, d, L, B = 9228, 128, 3, 1000
    ws = hip_lib.tpnet_stream_workspace_bytes(N, d, L, 64 * B, B)
    assert hip_lib.tpnet_stream_schedule(N, d, L, 20 * B, B, 0, ws) == 1
    assert hip_lib.tpnet_stream_schedule(N, d, L, 8 * B, B, 0, ws) == 0
    assert hip_lib.tpnet_stream_schedule(N, d, L, 8 * B, B, 16, ws) == 1          # TPNET_FLAG_SCHED_WINDOWED: from 4 batches
    assert hip_lib.tpnet_stream_schedule(N, d, L, 20 * B, B, 32, ws) == 0         # TPNET_FLAG_SCHED_BATCH
    assert hip_lib.tpnet_stream_schedule(N, d, L, 20 * B, B, 0, 1 << 20) == 0     # a workspace that holds no window
    assert hip_lib.tpnet_stream_schedule(N, 126, L, 20 * B, B, 0, ws) == 0        # rows that take no 16-byte vectors
    assert hip_lib.tpnet_stream_schedule(0, d, L, 20 * B, B, 0, ws) < 0


def test_a_launch_that_outlives_its_limit_is_ended_with_its_ranks(monkeypatch, tmp_path, capsys):
    """launch_ranks starts the launcher in a process group of its own and, past TPNET_BENCH_LAUNCH_TIMEOUT, ends that GROUP: a stand-in
    launcher that starts a 'rank' (a grandchild) and hangs must leave nothing behind -- a launcher killed alone would leave its ranks
    holding the GPUs."""
    import time
    import torch
    b = _bench()
    pidfile = tmp_path / "rank.pid"
    script = tmp_path / "hang.py"
    script.write_text(
        "import subprocess, sys, time\n"
        "p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
        f"open({str(pidfile)!r}, 'w').write(str(p.pid))\n"
        "time.sleep(600)\n")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(b, "launcher_command", lambda n, argv, port, python=None, script_=None: ([sys.executable, str(script)], dict(os.environ)))
    monkeypatch.setenv("TPNET_BENCH_LAUNCH_TIMEOUT", "3")
    monkeypatch.setenv("TPNET_BENCH_BACKEND", "gloo")
    t0 = time.time()
    rc = b.launch_ranks(2, ["--gpus", "2"])
    assert rc == 124 and time.time() - t0 < 40
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2
    rank_pid = int(pidfile.read_text())
    for _ in range(50):                                   # the grandchild is gone (or a zombie of init's about to be reaped)
        try:
            os.kill(rank_pid, 0)
        except ProcessLookupError:
            break
        try:
            if open(f"/proc/{rank_pid}/stat").read().split(")")[-1].split()[0] == "Z":
                break
        except FileNotFoundError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the stand-in rank survived its launcher")
