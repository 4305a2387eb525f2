"""Sanitizer builds of everything that runs without a GPU: the shared-memory transport's lock-free protocol
(museinference.jl_amd/csrc/shm_gather.hpp) under ThreadSanitizer -- with the ranks as THREADS sharing one mapping, the form
the sanitizer can follow -- and under AddressSanitizer + UndefinedBehaviorSanitizer with the ranks as processes; the CPU
oracle (oracle/muse_oracle.c) under ASan + UBSan on awkward shapes.  CPU only, never on the GPU box's device (GPU
AddressSanitizer is not available on this pool).  Not covered: muse_comm.cpp's RCCL worker thread, which cannot run
without a device; its hand-off is a mutex-protected queue plus one atomic flag per result area."""
import os
import subprocess
import uuid

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DRIVER = os.path.join(HERE, "native", "shm_gather_driver.cpp")
ORACLE_DRIVER = os.path.join(HERE, "native", "oracle_sanitize_driver.c")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")


def _build(tmp, name, cmd):
    exe = str(tmp / name)
    r = subprocess.run(cmd + ["-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip(f"sanitizer runtime not available for this build: {r.stderr[-400:]}")
    return exe


def _clean(proc_out, what):
    out = proc_out.stdout + proc_out.stderr
    assert proc_out.returncode == 0, f"{what}: exit {proc_out.returncode}\n{out[-3000:]}"
    for marker in ("ThreadSanitizer", "AddressSanitizer", "runtime error", "LeakSanitizer"):
        assert marker not in out, f"{what}: {marker} report\n{out[-3000:]}"


@pytest.mark.parametrize("nranks,rounds,block", [(2, 3000, 64), (4, 1500, 520), (8, 400, 1032)])
def test_shm_protocol_under_thread_sanitizer(tmp_path, nranks, rounds, block):
    exe = _build(tmp_path, "drv_tsan", ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", DRIVER, "-lrt", "-lpthread"])
    r = subprocess.run([exe, "--threads", "unused", str(nranks), str(rounds), str(block)], capture_output=True, text=True, env=ENV, timeout=600)
    _clean(r, f"TSan, {nranks} ranks as threads")
    assert r.stdout.count(" ok ") == nranks


@pytest.mark.parametrize("nranks", [2, 4])
def test_shm_protocol_under_address_and_ub_sanitizer(tmp_path, nranks):
    exe = _build(tmp_path, "drv_asan", ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                        DRIVER, "-lrt", "-lpthread"])
    name = "/muse_san_" + uuid.uuid4().hex[:16]
    procs = [subprocess.Popen([exe, name, str(nranks), str(r), "800", "257"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=ENV)
             for r in range(nranks)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        _clean(subprocess.CompletedProcess(p.args, p.returncode, out, err), f"ASan/UBSan, rank {r} of {nranks} processes")
    # ... and the ranks as threads of one process (every view through one mapping)
    r = subprocess.run([exe, "--threads", "unused", str(nranks), "500", "64"], capture_output=True, text=True, env=ENV, timeout=300)
    _clean(r, "ASan/UBSan, ranks as threads")


def test_step_algebra_under_address_and_ub_sanitizer(tmp_path):
    """csrc/step.hpp (the muse! step both native loops take on the host) with its bitwise self-check, sanitized."""
    exe = _build(tmp_path, "step_asan", ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                         "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                                         os.path.join(HERE, "native", "step_driver.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    _clean(r, "step.hpp under ASan/UBSan")
    assert "step driver ok" in r.stdout


def test_oracle_under_address_and_ub_sanitizer(tmp_path):
    exe = _build(tmp_path, "oracle_asan", ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fopenmp",
                                           "-mavx2", "-mfma", "-ffp-contract=off", ORACLE_DRIVER, os.path.join(ROOT, "oracle", "muse_oracle.c"), "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(ENV, OMP_NUM_THREADS="3"), timeout=600)
    _clean(r, "oracle under ASan/UBSan")
    assert "oracle sanitize driver ok" in r.stdout
