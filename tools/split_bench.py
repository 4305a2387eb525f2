"""Solver-launch time of a strongly scaled map's per-GPU share: nel elements at N, `split` workgroups per element."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import museinference_jl_amd as M
model, N, nth = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("funnel", 10000, 1)
theta = [1.0] * nth
for nel in (64, 128, 512):
    for split in (0, 2, 4, 8):
        p = M.HipMuseProblem(None, model=model, ntheta=nth, N=N)
        p.set_element_split(split)
        for _ in range(3):
            p.map_and_score_batch(0, 0, nel, theta)
        p.profile_begin(64)
        for k in range(50):
            p.map_and_score_batch_async(0, 0, nel, theta, result_area=k % 4)
            if k >= 3: p.batch_wait(nel, (k - 3) % 4)
        p.synchronize()
        ms = p.profile_end()
        t0 = time.perf_counter()
        for k in range(200):
            p.map_and_score_batch_async(0, 0, nel, theta, result_area=k % 4)
            if k >= 3: p.batch_wait(nel, (k - 3) % 4)
        p.synchronize()
        wall = (time.perf_counter() - t0) / 200
        print(f"{model} N={N} nel={nel} split={split}: kernel {1e3*np.median(ms):.1f} us (min {1e3*ms.min():.1f}), wall/step {1e6*wall:.1f} us", flush=True)
        p.close()
