"""`python bench.py --gpus N` must run by itself (the driver's command): bench.py's self-launcher starts N fresh ranks
before anything touches the GPU, relays rank 0's JSON line and returns the worst exit code.  On CPU: the launcher against
a stand-in rank that really rendezvouses over gloo; a failing rank; and bench.py itself, which must fail loudly (not
hang, not fall back) without a HIP device.  The full two-rank run on a GPU is tests/test_gpu_bench.py."""
import importlib.util
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "launcher_stub.py")


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_starts_ranks_and_relays_rank0_line(capfd):
    bench = load_bench()
    rc = bench.launch_ranks(STUB, ["--steps", "3"], 2)
    out, err = capfd.readouterr()
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                       # ONE line on stdout: rank 0's JSON
    got = json.loads(lines[0])
    assert got == {"world": 2, "sum": 3.0, "argv": ["--steps", "3"]}
    assert "banner line of rank 0" in err and "banner line of rank 1" in err   # everything else goes to stderr


def test_launcher_returns_the_worst_exit_code_and_does_not_hang(capfd):
    bench = load_bench()
    t0 = time.monotonic()
    rc = bench.launch_ranks(STUB, ["--fail-rank", "1"], 2, grace_s=2.0)   # rank 0 would wait in the rendezvous for minutes
    assert rc != 0
    assert time.monotonic() - t0 < 120
    out, _ = capfd.readouterr()
    assert out.strip() == ""


def test_bench_gpus2_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_bench.py runs the real thing")
    env = dict(os.environ, MUSE_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert p.stdout.strip() == ""                     # no result line
    assert "[rank" not in p.stdout
    assert "HIP" in p.stderr or "GPU" in p.stderr, p.stderr[-2000:]


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
