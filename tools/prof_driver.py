"""Diagnostic: where a complete muse(prob, ...; get_covariance=True) spends its wall time (cProfile)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
p0 = M.HipMuseProblem(None, model="funnel", N=10000)
x, _ = p0.sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0])
prob = M.HipMuseProblem(x, model="funnel", prior=M.GaussianPrior(0.0, 3.0))
for native in (True, False):
    M.muse(prob, [1.0], rng=0, nsims=512, get_covariance=True, native=native)
    t0 = time.perf_counter()
    for _ in range(20): r = M.muse(prob, [1.0], rng=0, nsims=512, get_covariance=True, native=native)
    print("native", native, "muse() wall per run: %.3f ms, iterations %d" % ((time.perf_counter() - t0) / 20 * 1e3, len(r.history)))
    t0 = time.perf_counter()
    for _ in range(20): r = M.muse(prob, [1.0], rng=0, nsims=512, maxsteps=30, theta_rtol=1e-9, native=native)
    print("native", native, "30 iterations, no covariance: %.3f ms per run (%.1f us per iteration)" % ((time.perf_counter() - t0) / 20 * 1e3, (time.perf_counter() - t0) / 20 / len(r.history) * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): M.muse(prob, [1.0], rng=0, nsims=512, get_covariance=True)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
