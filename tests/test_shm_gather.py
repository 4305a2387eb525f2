"""The shared-memory transport's protocol (museinference.jl_amd/csrc/shm_gather.hpp) on CPU: a small C++ driver
(tests/native/shm_gather_driver.cpp) is compiled with g++ and run as several processes."""
import os
import subprocess
import uuid

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "shm_gather_driver.cpp")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("shm") / "shm_gather_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, SRC, "-lrt", "-lpthread"])
    return exe


def run_ranks(exe, nranks, rounds, block, abort_at=None, extra=0):
    name = "/muse_test_" + uuid.uuid4().hex[:16]
    procs = []
    env = dict(os.environ, MUSE_TEST_EXTRA=str(extra)) if extra else None
    for r in range(nranks):
        cmd = [exe, name, str(nranks), str(r), str(rounds), str(block)] + ([str(abort_at)] if abort_at is not None else [])
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    out = [p.communicate(timeout=120) for p in procs]
    assert not os.path.exists("/dev/shm" + name), "the creator unlinks the segment once everyone is attached"
    return [p.returncode for p in procs], out


@pytest.mark.parametrize("nranks,rounds,block", [(2, 4000, 64), (4, 3000, 520), (8, 600, 4104)])
def test_every_rank_reads_every_block_of_every_round(driver, nranks, rounds, block):
    codes, out = run_ranks(driver, nranks, rounds, block)
    assert codes == [0] * nranks, out


def test_one_rank(driver):
    codes, out = run_ranks(driver, 1, 100, 8)
    assert codes == [0], out


def test_abort_releases_the_peers(driver):
    # the last rank gives up in round 37: the others leave their wait with "aborted by a peer" instead of a timeout
    codes, out = run_ranks(driver, 3, 1000, 64, abort_at=37)
    assert codes[-1] == 43 and codes[:-1] == [42, 42], (codes, out)


@pytest.mark.parametrize("nranks,block,extra", [(3, 64, 256 * 1024), (8, 520, 5000)])
def test_extra_region_behind_the_blocks(driver, nranks, block, extra):
    """The region the sharded device loop's score board lives in (muse_comm.cpp registers it with the HIP runtime): on a page
    boundary of its own, a whole number of pages, zero-filled, and the same memory in every rank."""
    codes, out = run_ranks(driver, nranks, 200, block, extra=extra)
    assert codes == [0] * nranks, out
