import cProfile, pstats, io, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import museinference_jl_amd as M
N, nsims = 10000, 512
xdata, _ = M.HipMuseProblem(None, model="funnel", ntheta=1, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0])
prob = M.HipMuseProblem(xdata, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
for _ in range(3):
    M.muse(prob, [1.0], rng=0, nsims=nsims, get_covariance=True)
t0 = time.perf_counter()
for _ in range(20):
    r = M.muse(prob, [1.0], rng=0, nsims=nsims, get_covariance=True)
print("muse(get_covariance=True): %.1f us per call, %d iterations" % (1e6 * (time.perf_counter() - t0) / 20, len(r.history)))
t0 = time.perf_counter()
for _ in range(20):
    r = M.muse(prob, [1.0], rng=0, nsims=nsims)
print("muse(): %.1f us per call" % (1e6 * (time.perf_counter() - t0) / 20))
t0 = time.perf_counter()
for _ in range(20):
    prob.run_muse(0, [1.0], nsims=nsims, maxsteps=50, theta_rtol=0.1, atol=1e-2, alpha=0.7)
print("run_muse alone: %.1f us per call" % (1e6 * (time.perf_counter() - t0) / 20))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    M.muse(prob, [1.0], rng=0, nsims=nsims, get_covariance=True)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
