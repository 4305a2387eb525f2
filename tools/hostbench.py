import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import museinference_jl_amd as M
p = M.HipMuseProblem(None, model="funnel", ntheta=1, N=10000)
nel=512; theta=[1.0]
for _ in range(3): p.map_and_score_batch(0, 0, nel, theta)
for rep in range(3):
    te = 0.0
    t0 = time.perf_counter()
    for k in range(300):
        a = time.perf_counter()
        p.map_and_score_batch_async(0, 0, nel, theta, result_area=k % 4)
        te += time.perf_counter() - a
        if k >= 3: p.batch_wait(nel, (k - 3) % 4)
    p.synchronize()
    wall = (time.perf_counter() - t0) / 300
    print(f"{os.environ.get('TAG','')} wall/step {1e6*wall:.1f} us, enqueue {1e6*te/300:.1f} us", flush=True)
