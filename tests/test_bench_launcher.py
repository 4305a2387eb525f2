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


def test_flat_scalars_of_the_sharded_extras_survive_every_outcome():
    """bench.py's N > 1 extras reach the driver's record as FLAT scalars of `config` (flat_sharded): with every part measured, with a
    part skipped (an exception inside it), with everything skipped -- never a KeyError between the measurement and the line; every
    value is a scalar the driver's parser keeps."""
    bench = load_bench()
    run = lambda loop, us: {"us_per_outer_iteration_30": us, "us_per_outer_iteration_steady": us - 1.0, "iterations": 30, "loop_ran": loop, "by_regime": {}}
    hs = {"board": "device", "device_handshake": 1, "host_handshake": 1, "device_seen": 255, "host_seen": 255, "last_loop": "none",
          "device_wait_us": 12.5, "host_wait_us": 30.0}
    full = {"muse_run": {"nsims": 512, "N": 10000, "ranks_seen": 8, "board": "device", "handshake": hs, "board_setup_ms": 3.2,
                         "runs": {"default": run("device", 17.0), "host_board": run("host", 21.0), "host_loop": run("none", 25.0)},
                         "trajectory_bit_equal_to_unsharded_on_every_rank": True, "theta": [0.1]},
            "cfg4_fd_H": {"ms_per_call": 0.09, "problems_per_call": 4097, "maps_per_s": 1.0, "converged_on_every_rank": True, "units_per_rank": 256, "nsims": 512},
            "cfg5_smooth_1e5": {"ms_per_step": 2.3, "sims_per_s": 1.0, "nsims": 1024, "N": 100000, "sims_per_rank": 128, "converged_on_every_rank": True}}
    cfg = {}
    bench.flat_sharded(cfg, full)
    assert cfg["sharded_muse_iter_us"] == 17.0 and cfg["sharded_loop_ran"] == "device" and cfg["sharded_bit_equal"] is True
    assert cfg["handshake_device"] == 1 and cfg["handshake_host"] == 1 and cfg["sharded_ranks_seen"] == 8
    assert cfg["cfg4_fd_H_sharded_ms"] == 0.09 and cfg["cfg5_smooth_1e5_sharded_ms"] == 2.3 and cfg["cfg4_fd_H_sharded_ok"] is True
    assert all(isinstance(v, (int, float, str, bool)) or v is None for v in cfg.values()), cfg
    partly = dict(full, cfg4_fd_H={"skipped": "MuseError: x"}, muse_run={"skipped": "RuntimeError: y"})
    cfg = {}
    bench.flat_sharded(cfg, partly)
    assert cfg["sharded_muse_skipped"] == "RuntimeError: y" and cfg["cfg4_fd_H_sharded_skipped"] == "MuseError: x" and cfg["cfg5_smooth_1e5_sharded_ms"] == 2.3
    cfg = {}
    bench.flat_sharded(cfg, {})
    assert cfg["sharded_muse_skipped"] == "not run" and cfg["cfg4_fd_H_sharded_skipped"] == "not run"
    json.dumps(cfg)


def test_flat_scalars_of_the_single_gpu_extras_survive_missing_parts():
    bench = load_bench()
    out = {"config": {}, "extra": {"workloads": {"noise_1e6": {"skipped": "x"}, "funnel4_1e4": {"ms_per_step": 0.058, "frac": 0.4, "bound": "valu", "frac_kernel_time": 0.39}},
                                   "muse_run": {"skipped": "x"}, "muse_run_8gpu_share": {"skipped": "y"}, "scale_projection": {"cfg4_fd_H": {"skipped": "z"}}}}
    bench.flat_single(out)
    assert out["config"]["funnel4_ms"] == 0.058 and "noise_1e6_ms" not in out["config"] and "proj_cfg4" not in out["config"]
    bench.flat_single({"config": {}})    # (--no-extra: nothing to flatten)
