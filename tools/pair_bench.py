"""Per-step time of a 512-sim cold map at N = 10^4 for the two-parameter user-model family (models/normal_mean_var.h) next to the
built-in funnel with the same number of theta components.  python tools/pair_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import museinference_jl_amd as M

def run(model, nth, theta, N=10000, nsims=512, steps=200, placement=-1):
    prob = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
    if placement >= 0:
        prob.set_placement(placement)
    prob.set_normals_cache(False)
    prob.set_concurrency(2)
    outs = [(np.empty((nsims, nth)), np.zeros(nsims, dtype=M._capi.INFO_DTYPE)) for _ in range(4)]
    def go(K):
        for k in range(K):
            if k >= 3:
                prob.batch_wait(nsims, (k + 1) % 4, out=outs[(k + 1) % 4])
            prob.map_and_score_batch_async(0, 0, nsims, theta, atol=1e-2, z0_mode=0, result_area=k % 4)
        for k in range(max(0, K - 3), K):
            prob.batch_wait(nsims, k % 4, out=outs[k % 4])
    go(20)
    prob.synchronize()
    t0 = time.perf_counter()
    go(steps)
    prob.synchronize()
    dt = (time.perf_counter() - t0) / steps
    info = outs[(steps - 1) % 4][1]
    prob.close()
    return 1e6 * dt, float(info["f_calls"].mean()), float(info["iterations"].mean())

pair = M.ElementwiseModel.packaged("normal_mean_var")
for nth in (2, 4, 8):
    th = [0.3] * (nth // 2) + [1.0] * (nth // 2)
    print("pair   ntheta", nth, "us/step %.1f  f_calls %.2f iterations %.2f" % run(pair, nth, th))
    print("pair   ntheta", nth, "streaming placement: us/step %.1f  f_calls %.2f iterations %.2f" % run(pair, nth, th, placement=0))
for nth in (1, 2, 4, 8):
    print("funnel ntheta", nth, "us/step %.1f  f_calls %.2f iterations %.2f" % run("funnel", nth, [1.0] * nth))
