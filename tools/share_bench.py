"""The 8-GPU share of a DEPENDENT map sequence on one GPU: the muse! loop (src/muse.jl:159-232) at nsims/8 + 1 elements --
what ONE rank of an 8-GPU job runs per outer iteration -- timed three ways: the sharded native loop (muse_run_sharded: gathered
map through the shared-memory transport with this one rank, step on the host), and for comparison the unsharded native loops
on the same 64 + 1 elements (no exchange at all).  python tools/share_bench.py [N] [nsims_total] [ngpus] [split]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import museinference_jl_amd as M

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
SPLIT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
share = S // G


def measure(nsims=share, n_iter=30, split=SPLIT):
    xdata, _ = M.HipMuseProblem(None, model="funnel", ntheta=1, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0])
    prob = M.HipMuseProblem(xdata, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    if split:
        prob.set_element_split(split)
    kw = dict(nsims=nsims, maxsteps=n_iter, theta_rtol=1e-12, atol=1e-2, alpha=0.7)
    out = {}
    for name, fn in (("host_loop", lambda: prob.run_muse(0, [1.0], device_loop=False, **kw)),
                     ("device_loop", lambda: prob.run_muse(0, [1.0], device_loop=True, **kw))):
        best, dev = 1e9, 0.0
        for _ in range(5):
            t0 = time.perf_counter()
            n, theta, hist, gs, info = fn()
            best = min(best, (time.perf_counter() - t0) / n)
            dev = float(np.median(hist[5:, -1]))
        out[name] = {"us_per_iteration_wall": 1e6 * best, "us_per_iteration_steady": 1e6 * dev}
    prob.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm", 4096))
    best, dev = 1e9, 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        n, theta, hist, gs, info = prob.run_muse_sharded(0, [1.0], **kw)
        best = min(best, (time.perf_counter() - t0) / n)
        dev = float(np.median(hist[5:, -1]))
    out["sharded_shm_1rank"] = {"us_per_iteration_wall": 1e6 * best, "us_per_iteration_steady": 1e6 * dev}
    prob.close()
    return out


if __name__ == "__main__":
    import json
    one = measure(nsims=S)            # the whole job on this GPU, for the ratio
    part = measure(nsims=share)
    print(json.dumps({"N": N, "nsims_total": S, "ngpus": G, "element_split": SPLIT, "whole_job_one_gpu": one, "share_of_one_rank": part,
                      "projected_speedup_steady": {k: one["device_loop"]["us_per_iteration_steady"] / part[k]["us_per_iteration_steady"]
                                                   for k in part}}, indent=1))
