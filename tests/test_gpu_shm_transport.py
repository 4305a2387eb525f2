"""The shared-memory transport of the sharded map (muse_comm_unique_id_ex(MUSE_TRANSPORT_SHM), the pool seam of
src/util.jl:74-83 for the ranks of one node) on a GPU: several processes -- all on GPU 0, the transport has no device
side -- run the pipelined gathered map and the synchronous collectives; every rank must hold, bit for bit, what one
process computes for the whole sim range."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [1, 3])
def test_gathered_map_and_collectives_over_shared_memory(world, tmp_path):
    import museinference_jl_amd as M
    uid = M.HipMuseProblem.comm_unique_id("shm", 4096)   # naming the segment does not touch the GPU
    uid2 = M.HipMuseProblem.comm_unique_id("shm", 4096)  # (the communicator of the native sharded muse! loop)
    uid3 = M.HipMuseProblem.comm_unique_id("shm", 4096)  # (the 12-component problem's)
    uid4 = M.HipMuseProblem.comm_unique_id("shm", 4096)  # (the headline-shaped sharded loop's)
    assert len(uid) == 128
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    env = dict(os.environ, MUSE_DEBUG_RUN_TIMING="1")     # (the loop kernel's launches leave a line on stderr: which loop ran)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shm_rank_worker.py"), uid.hex(), str(world), str(r), outs[r], uid2.hex(), uid3.hex(),
                               uid4.hex()], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert [p.returncode for p in procs] == [0] * world, "\n".join(logs)
    res = [np.load(o) for o in outs]

    NSIMS, N, NTH, SEED = 23, 2000, 2, 77
    rows = -(-NSIMS // world)
    ref = M.HipMuseProblem(None, model="funnel", ntheta=NTH, N=N)
    for k in range(9):
        th = np.array([0.3 * k - 1.0, 0.5 - 0.1 * k])
        g, info = ref.map_and_score_batch(SEED, 0, NSIMS, th, atol=1e-4)
        for r in range(world):
            g_all = res[r][f"g{k}"]
            assert g_all.shape == (world, rows, NTH)
            for q in range(world):
                lo, hi = M.block_partition(0, NSIMS, world, q)
                assert np.array_equal(g_all[q, : hi - lo], g[lo:hi]), (k, r, q)
                assert np.all(g_all[q, hi - lo:] == 0.0)           # padding rows of a short block
            lo, hi = M.block_partition(0, NSIMS, world, r)
            assert np.array_equal(res[r][f"it{k}"], info["iterations"][lo:hi])
    ref.close()
    refb = M.HipMuseProblem(None, model="funnel", ntheta=12, N=N)      # the big tier's score blocks travel like the others
    gb, ib = refb.map_and_score_batch(SEED, 0, NSIMS, np.linspace(-0.5, 0.6, 12), atol=1e-4)
    refb.close()
    for r in range(world):
        assert res[r]["gbig"].shape == (world, rows, 12)
        for q in range(world):
            lo, hi = M.block_partition(0, NSIMS, world, q)
            assert np.array_equal(res[r]["gbig"][q, : hi - lo], gb[lo:hi]), (r, q)
        lo, hi = M.block_partition(0, NSIMS, world, r)
        assert np.array_equal(res[r]["itbig"], ib["iterations"][lo:hi])
    # muse_run_sharded on every rank = the unsharded native loop, bit for bit (theta is never exchanged: every rank takes the
    # same step from the same gathered scores)
    xdat = np.sin(0.37 * np.arange(N)) * 1.3
    one = M.HipMuseProblem(xdat, model="funnel", ntheta=NTH, prior=M.GaussianPrior(0.0, 3.0))
    n1, t1, h1, g1, i1 = one.run_muse(SEED, [1.0, 0.4], nsims=NSIMS, maxsteps=6, theta_rtol=0.0, atol=1e-3, alpha=0.7, device_loop=False)
    one.close()
    assert n1 == 6
    for r in range(world):
        assert int(res[r]["run_n"]) == n1 and np.array_equal(res[r]["run_theta"], t1)
        assert np.array_equal(res[r]["run_hist"], h1[:, :-1]) and np.array_equal(res[r]["run_gs"], g1)
        lo, hi = M.block_partition(0, NSIMS, world, r)
        mine = np.concatenate([i1["iterations"][:, :1], i1["iterations"][:, 1 + lo:1 + hi]], axis=1) if r == 0 else i1["iterations"][:, 1 + lo:1 + hi]
        assert np.array_equal(res[r]["run_it"], mine)
    # The sharded loop at the headline's shape, as ONE persistent launch per rank -- the ranks' kernels resident side by side on this
    # one GPU, their scores meeting on the node's board in pinned host memory -- and as the host-driven loop: both the unsharded
    # loop's trajectory, bit for bit, on every rank; and the persistent launch is what ran (its timing line on stderr: the first
    # sharded loop above -- N = 2000, two components, one element per worker -- and four of the six runs here: the boards in device
    # memory, mapped across the processes by hipIpc, and the one board in pinned host memory).
    xd1 = np.cos(0.11 * np.arange(10000)) * 1.7
    one = M.HipMuseProblem(xd1, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    kw = dict(nsims=64, maxsteps=8, theta_rtol=0.0, atol=1e-2, alpha=0.7)
    n1, t1, h1, g1, i1 = one.run_muse(SEED, [1.0], device_loop=False, **kw)
    n2, t2, h2, g2, i2 = one.run_muse(SEED, t1, device_loop=False, z0_warm=True, **dict(kw, maxsteps=3))
    one.close()
    for r in range(world):
        lo, hi = M.block_partition(0, 64, world, r)
        for tag in ("dev", "hostboard", "host"):
            assert int(res[r][f"s1_{tag}_n"]) == n1 and np.array_equal(res[r][f"s1_{tag}_theta"], t1), (r, tag)
            assert np.array_equal(res[r][f"s1_{tag}_hist"], h1[:, :-1]) and np.array_equal(res[r][f"s1_{tag}_gs"], g1), (r, tag)
            mine = np.concatenate([i1[:, :1], i1[:, 1 + lo:1 + hi]], axis=1) if r == 0 else i1[:, 1 + lo:1 + hi]
            assert np.array_equal(res[r][f"s1_{tag}_it"], mine["iterations"]) and np.array_equal(res[r][f"s1_{tag}_fc"], mine["f_calls"]), (r, tag)
            assert np.array_equal(res[r][f"s1_{tag}_warm_theta"], t2) and np.array_equal(res[r][f"s1_{tag}_warm_gs"], g2), (r, tag)
        # which loop each run took, from the library's own record (muse_comm_board_status) ...
        assert [str(res[r][f"s1_{tag}_last_loop"]) for tag in ("dev", "hostboard", "host")] == ["device", "host", "none"], r
        # ... and the verdict of the boards' set-up hand-shake (round 6): both kinds of board were PROVED before the first loop used
        # them -- every rank's one-wavefront kernel stored its tagged pair into every rank's board and saw every peer's pair in its own
        # within the millisecond bound; the persistent loop uses the boards in device memory
        bs = res[r]["board_status"]
        assert bs[0] == 2 and bs[1] == 1 and bs[2] == 1, (r, bs)
        assert bs[3] == (1 << world) - 1 and bs[4] == (1 << world) - 1, (r, bs)
        assert np.all(res[r]["board_wait_us"] < 50e3), res[r]["board_wait_us"]
        assert logs[r].count("board hand-shake -- device boards ok") == 2, logs[r]   # (the two contexts that ran sharded loops)
        assert logs[r].count("[muse_run_device]") == 5, logs[r]
        # ... through the boards in device memory (the first sharded loop and two runs here), the host board (two), the host loop (two)
        assert logs[r].count("boards in device memory (hipIpc)") == 3 and logs[r].count("board in pinned host memory") == 2, logs[r]
        assert logs[r].count("host-driven loop") == 2, logs[r]
    big = [np.sin(np.arange(40000.0) * (q + 1)) for q in range(world)]
    total = big[0].copy()
    for q in range(1, world):
        total = total + big[q]                                      # rank order: bitwise the same on every rank
    for r in range(world):
        assert np.array_equal(res[r]["ag"], np.stack([np.arange(5.0) + 10.0 * q for q in range(world)]))
        assert np.array_equal(res[r]["ar"], total)
        assert np.array_equal(res[r]["ag_big"], np.stack([b[:20001] for b in big]))


def test_block_capacity_is_checked():
    import museinference_jl_amd as M
    prob = M.HipMuseProblem(None, model="funnel", ntheta=2, N=600)
    prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", 8))
    with pytest.raises(M.MuseError, match="capacity"):
        prob.map_and_score_batch_gather_async(1, 0, 5, np.zeros(2), 5)      # 5 rows x 2 > 8 doubles
    n = prob.map_and_score_batch_gather_async(1, 0, 4, np.zeros(2), 4)
    g_all, info = prob.batch_wait_gathered(n, 4)
    g, _ = prob.map_and_score_batch(1, 0, 4, np.zeros(2))
    assert np.array_equal(g_all[0], g)
    prob.close()


@pytest.mark.parametrize("env,board,dev_hs,host_hs", [
    ({}, "device", 1, 1), ({"MUSE_DEBUG_NO_IPC_BOARD": "1"}, "host", -1, 1), ({"MUSE_DEBUG_NO_BOARD": "1"}, "none", -1, -1),
    # the failure the hand-shake exists for -- granules stored into a board never become visible to the GPU that polls it (test hook:
    # they are stored beside the slots) --: that kind of board is dropped on every rank after the (here 3-ms) bound, the next kind is
    # used, and the answer is the same bits
    ({"MUSE_DEBUG_HANDSHAKE_FAIL": "1", "MUSE_BOARD_HANDSHAKE_MS": "3"}, "host", 0, 1),
    ({"MUSE_DEBUG_HANDSHAKE_FAIL": "3", "MUSE_BOARD_HANDSHAKE_MS": "3"}, "none", 0, 0)])
def test_board_status_follows_the_switches_and_every_board_gives_the_same_bits(env, board, dev_hs, host_hs):
    """muse_comm_board_status on a one-rank shared-memory communicator under the set-up switches (read at context creation, so in a
    process of its own): the board the persistent loop will use, the hand-shake's verdict per kind of board (-1: not tried), which
    loop the run then took -- and the trajectory is the unsharded loop's bit for bit whichever it was.  An RCCL communicator has no
    boards at all."""
    code = r'''
import sys, json
import numpy as np
import museinference_jl_amd as M
x = np.cos(0.11 * np.arange(10000)) * 1.7
kw = dict(nsims=40, maxsteps=6, theta_rtol=0.0, atol=1e-2, alpha=0.7)
one = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
a = one.run_muse(5, [1.0], device_loop=False, **kw)
one.close()
p = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
p.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", 4096))
before = p.comm_board_status()
b = p.run_muse_sharded(5, [1.0], **kw)
after = p.comm_board_status()
p.close()
same = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2][:, :-1], b[2][:, :-1]) and np.array_equal(a[3], b[3])
r = M.HipMuseProblem(None, model="funnel", ntheta=1, N=1000)
r.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("rccl"))
rc = r.comm_board_status()
r.close()
print(json.dumps({"before": before, "after": after, "same": bool(same), "rccl": rc}))
'''
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(HERE))
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["same"] is True
    assert d["before"]["board"] == board and d["before"]["device_handshake"] == dev_hs and d["before"]["host_handshake"] == host_hs, d
    assert d["before"]["last_loop"] == "none" and d["after"]["last_loop"] == board, d
    assert d["rccl"]["board"] == "none" and d["rccl"]["device_handshake"] == -1 and d["rccl"]["host_handshake"] == -1
    if dev_hs == 0:      # a failed hand-shake waited its bound (3 ms), not the loop kernel's 4 s
        assert 2.9e3 <= d["before"]["device_wait_us"] <= 10e3 and d["before"]["device_seen"] == 0, d["before"]
