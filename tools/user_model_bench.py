"""Throughput of a user-supplied model (include/muse_model.h) next to the built-in funnel at BASELINE.json configs[1]'s shape
(N = 10^4, 512 sims per step, pipelined over result areas and two lanes as bench.py does): the packaged `cubic` model (a
non-quadratic MAP objective: several L-BFGS iterations per simulation) and the funnel written as a user's header (the same
bits as the built-in one -- and the same speed, if the seam costs nothing).  Development aid; prints one line per model.
Usage (GPU box): [THETA=0.0] python tools/user_model_bench.py [steps] [only the cases whose label contains this]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import museinference_jl_amd as M

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N, nsims, AREAS = 10000, 512, 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THETA = float(os.environ.get("THETA", "1.0"))
cases = [("funnel (built in)", "funnel", 1e-2), ("gaussian_funnel (header)", M.ElementwiseModel.packaged("gaussian_funnel"), 1e-2),
         ("cubic", M.ElementwiseModel.packaged("cubic"), 1e-2), ("cubic atol 1e-6", M.ElementwiseModel.packaged("cubic"), 1e-6)]
only = sys.argv[2] if len(sys.argv) > 2 else ""
for label, model, atol in cases:
    if only not in label:
        continue
    prob = M.HipMuseProblem(None, model=model, ntheta=1, N=N)
    prob.set_concurrency(2)
    outs = [(np.empty((nsims, 1)), np.zeros(nsims, dtype=M._capi.INFO_DTYPE)) for _ in range(AREAS)]

    def run(K):
        for k in range(K):
            if k >= AREAS - 1:
                prob.batch_wait(nsims, (k + 1) % AREAS, out=outs[(k + 1) % AREAS])
            prob.map_and_score_batch_async(0, k * nsims, (k + 1) * nsims, [THETA], atol=atol, z0_mode=0, result_area=k % AREAS)
        for k in range(max(0, K - AREAS + 1), K):
            prob.batch_wait(nsims, k % AREAS, out=outs[k % AREAS])

    run(20)
    prob.synchronize()
    t0 = time.perf_counter()
    run(steps)
    prob.synchronize()
    dt = time.perf_counter() - t0
    info = outs[(steps - 1) % AREAS][1]
    # the resident placement keeps x, z, s, g on chip; the (dx, dg) history of L-BFGS lives in HBM: per iteration k the two-loop
    # recursion reads its h_k pairs twice and the update writes one pair -> (4 sum h_k + 2 K) words of N doubles per sim, + zhat
    words = 4 * info["hist_words"].astype(np.int64) + 2 * info["iterations"].astype(np.int64) + 1
    tbs = 8.0 * N * words.sum() / (dt / steps) / 1e12
    print(f"{label:20s} {dt / steps * 1e6:8.1f} us per 512-sim step  {nsims * steps / dt / 1e6:7.3f} M sims/s   "
          f"iterations {info['iterations'].mean():.2f}  evaluations {info['f_calls'].mean():.2f}  status max {info['status'].max()}  "
          f"history traffic {tbs:.2f} TB/s ({8.0 * N * words.sum() / 1e9:.3f} GB per launch)")
    prob.close()
