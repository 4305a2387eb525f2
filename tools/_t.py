import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import museinference_jl_amd as M
N, nsims, nth = 10000, 512, 1
xdata, _ = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0] * nth)
prob = M.HipMuseProblem(xdata, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
for ms in (30, 30, 30, 1, 1, 2, 2, 60, 60):
    t0 = time.perf_counter()
    n = prob.run_muse(0, [1.0], nsims=nsims, maxsteps=ms, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=True)[0]
    print(ms, n, "wall us", 1e6 * (time.perf_counter() - t0), flush=True)
