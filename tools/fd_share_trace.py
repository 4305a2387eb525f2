"""Diagnostic: one rank's share of configs[3]'s finite-difference get_H! (bench.py: cfg4_fd_H share) called repeatedly -- for
`rocprofv3 --kernel-trace` (start/end stamps of the call's kernels: where its ~73 us go) -- with the host's wall time per call.
    python tools/fd_share_trace.py [calls] [sharded 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import museinference_jl_amd as M
import bench

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sharded = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = bench.FD_WORKLOAD
nunits = w["nsims"] * w["ntheta"]
prob = M.HipMuseProblem(None, model=w["model"], ntheta=w["ntheta"], N=w["N"])
lo, hi = M.block_partition(0, nunits, 8, 0)
if sharded:
    prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", 4096))
bench.fd_call(M, prob, 0, lo, hi, bool(sharded))
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    bench.fd_call(M, prob, 0, lo, hi, bool(sharded))
    ts.append(time.perf_counter() - t0)
print(f"share call ({2 * (hi - lo) + 1} problems, sharded={sharded}): min {1e6 * min(ts):.1f} us  median {1e6 * float(np.median(ts)):.1f} us")
prob.close()
