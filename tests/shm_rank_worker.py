"""One rank of tests/test_gpu_shm_transport.py: python shm_rank_worker.py <id hex> <nranks> <rank> <out.npz> <second id hex> <third id hex> <fourth id hex>.
All ranks share GPU 0 (the shared-memory transport has no device-side part)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import museinference_jl_amd as M

uid, world, rank, out = bytes.fromhex(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
NSIMS, N, NTH, SEED = 23, 2000, 2, 77
prob = M.HipMuseProblem(None, model="funnel", ntheta=NTH, N=N)
prob.comm_init(world, rank, uid)
assert prob.comm_transport() == "shm"
lo, hi = M.block_partition(0, NSIMS, world, rank)
rows = -(-NSIMS // world)
res = {}
# a pipelined sequence over the four result areas, a different theta per step
thetas = [np.array([0.3 * k - 1.0, 0.5 - 0.1 * k]) for k in range(9)]
pending = []
for k, th in enumerate(thetas):
    n = prob.map_and_score_batch_gather_async(SEED, lo, hi, th, rows, atol=1e-4, result_area=k % 4)
    pending.append((k, n))
    if len(pending) > 3:
        kk, nn = pending.pop(0)
        g_all, info = prob.batch_wait_gathered(nn, rows, kk % 4)
        res[f"g{kk}"], res[f"it{kk}"] = g_all, info["iterations"]
while pending:
    kk, nn = pending.pop(0)
    g_all, info = prob.batch_wait_gathered(nn, rows, kk % 4)
    res[f"g{kk}"], res[f"it{kk}"] = g_all, info["iterations"]
# synchronous collectives; the second message is longer than one block of the segment (moves in pieces)
res["ag"] = prob.allgather_scores(np.arange(5.0) + 10.0 * rank)
big = np.sin(np.arange(40000.0) * (rank + 1))
res["ar"] = prob.allreduce_sum(big)
res["ag_big"] = prob.allgather_scores(big[:20001])
prob.close()
# the big tier (12 components > MUSE_MAX_THETA) through the same gathered map
pb = M.HipMuseProblem(None, model="funnel", ntheta=12, N=N)
pb.comm_init(world, rank, bytes.fromhex(sys.argv[6]))
nb = pb.map_and_score_batch_gather_async(SEED, lo, hi, np.linspace(-0.5, 0.6, 12), rows, atol=1e-4, result_area=1)
res["gbig"], ib = pb.batch_wait_gathered(nb, rows, 1)
res["itbig"] = ib["iterations"]
pb.close()
# the muse! loop over the ranks in native code (muse_run_sharded): the same theta trajectory, records and scores on every rank
xdat = np.sin(0.37 * np.arange(N)) * 1.3
pm = M.HipMuseProblem(xdat, model="funnel", ntheta=NTH, prior=M.GaussianPrior(0.0, 3.0))
pm.comm_init(world, rank, bytes.fromhex(sys.argv[5]))
n, theta, hist, gs, info = pm.run_muse_sharded(SEED, [1.0, 0.4], nsims=NSIMS, maxsteps=6, theta_rtol=0.0, atol=1e-3, alpha=0.7)
res["run_n"], res["run_theta"], res["run_hist"], res["run_gs"], res["run_it"] = n, theta, hist[:, :-1], gs, info["iterations"]
pm.close()
# ... and at the headline's shape (N = 10^4, one component: the LDS-resident placement, the MAP kept in registers from one iteration
# to the next), by both loops: the persistent launch per rank whose scores meet on the node's board (round 5: what the library
# runs), and the host-driven loop (debug flag bit 17).  A second run continues from the resident MAPs (z0_warm).
xd1 = np.cos(0.11 * np.arange(10000)) * 1.7
p1 = M.HipMuseProblem(xd1, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
p1.comm_init(world, rank, bytes.fromhex(sys.argv[7]))
kw = dict(nsims=64, maxsteps=8, theta_rtol=0.0, atol=1e-2, alpha=0.7)
res["board_status"] = np.array([-9] * 6)
for tag, flags in (("dev", 0), ("hostboard", M.HipMuseProblem.DEBUG_HOST_BOARD), ("host", M.HipMuseProblem.DEBUG_SHARDED_HOST_LOOP)):
    # dev: a board per GPU in device memory, every rank's mapped into every rank (hipIpc); hostboard: the one board in pinned host
    # memory; host: the host-driven loop -- chosen on the live context (debug_flags: the environment is read once, at context creation)
    p1.debug_flags(flags | M.HipMuseProblem.DEBUG_RUN_TIMING)
    n, theta, hist, gs, info = p1.run_muse_sharded(SEED, [1.0], **kw)
    res[f"s1_{tag}_n"], res[f"s1_{tag}_theta"], res[f"s1_{tag}_hist"], res[f"s1_{tag}_gs"] = n, theta, hist[:, :-1], gs
    res[f"s1_{tag}_it"], res[f"s1_{tag}_fc"] = info["iterations"], info["f_calls"]
    res[f"s1_{tag}_last_loop"] = np.array(p1.comm_board_status()["last_loop"])
    n, theta, hist, gs, info = p1.run_muse_sharded(SEED, theta, z0_warm=True, **dict(kw, maxsteps=3))
    res[f"s1_{tag}_warm_theta"], res[f"s1_{tag}_warm_gs"] = theta, gs
p1.debug_flags(0)
bs = p1.comm_board_status()   # the set-up hand-shake's verdict for both kinds of board (made by the first sharded run above)
res["board_status"] = np.array([{"none": 0, "host": 1, "device": 2}[bs["board"]], bs["device_handshake"], bs["host_handshake"], bs["device_seen"],
                                bs["host_seen"], 0])
res["board_wait_us"] = np.array([bs["device_wait_us"], bs["host_wait_us"]])
p1.close()
np.savez(out, **res)
