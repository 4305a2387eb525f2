"""One rank of tests/test_gpu_shm_transport.py: python shm_rank_worker.py <id hex> <nranks> <rank> <out.npz>.
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
np.savez(out, **res)
