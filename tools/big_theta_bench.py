"""Wall time per 512-simulation map of the funnel at N = 10^4 against the number of theta components: the small tiers in their
LDS-resident placement and forced into the streaming one, and the big tier (ntheta > MUSE_MAX_THETA: streaming only, per-block
tables from the kernel-argument segment, block sums eight at a time).  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import museinference_jl_amd as M  # noqa: E402


def run(nth, placement, N=10000, nsims=512, reps=20):
    prob = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    prob.set_normals_cache(False)
    if placement is not None:
        prob.set_placement(placement)
    theta = np.linspace(-0.5, 1.0, nth)
    for _ in range(3):
        prob.map_and_score_batch(1, 0, nsims, theta, atol=1e-2, z0_mode=0)
    t0 = time.perf_counter()
    for _ in range(reps):
        g, info = prob.map_and_score_batch(1, 0, nsims, theta, atol=1e-2, z0_mode=0)
    dt = (time.perf_counter() - t0) / reps
    prob.close()
    return {"ntheta": nth, "placement": "auto" if placement is None else "streaming", "us_per_map": round(dt * 1e6, 1),
            "iterations_mean": float(np.mean(info["iterations"]))}


if __name__ == "__main__":
    rows = [run(1, None), run(8, None), run(8, 0), run(9, None), run(16, None), run(32, None), run(64, None)]
    print(json.dumps({"workload": "funnel N=10000, 512 sims per map, atol=1e-2, cold start", "rows": rows}))
