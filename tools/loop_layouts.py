"""The loop kernel's three deals of a job with more elements than workgroups, side by side on one box (production library): per-iteration
times of a 30-iteration muse_run_device call at configs[1], by regime (bench.iteration_regimes), and the call's wall per iteration.
    python tools/loop_layouts.py [N] [nsims] [ntheta]
flags (muse_debug_flags): 0 the stepper owns elements, the data element dealt with the rest (default); 64 the data element is the
stepper's; 128 the stepper only steps (the layout before); 256 the default deal without the data vector's trip through the g area."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import museinference_jl_amd as M
import bench
N, nsims, nth = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 512, 1)
xdata, _ = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0] * nth)
prob = M.HipMuseProblem(xdata, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
lib = M.load_library()
for rep in range(2):
    for flags, name in ((0, "stepper solves, data dealt"), (256, "... data not sent ahead"), (64, "stepper solves, owns data"), (128, "stepper only steps")):
        lib.muse_debug_flags(prob._ctx, flags)
        best, regs = 1e9, []
        for _ in range(5):
            t0 = time.perf_counter()
            n, theta, hist, gs, info = prob.run_muse(0, [1.0] * nth, nsims=nsims, maxsteps=30, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=True)
            best = min(best, (time.perf_counter() - t0) / n)
            regs.append(bench.iteration_regimes(hist, info))
        ls = np.median([r["line_search"]["us_per_outer_iteration"] for r in regs if "line_search" in r])
        cv = np.median([r["converged_at_start"]["us_per_outer_iteration"] for r in regs if "converged_at_start" in r])
        # a run as short as muse()'s own (the reference's default theta_rtol stops configs[1] after 2-3 iterations)
        t0 = time.perf_counter()
        for _ in range(20):
            n3 = prob.run_muse(0, [1.0] * nth, nsims=nsims, maxsteps=3, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=True)[0]
        short = (time.perf_counter() - t0) / 20
        print(f"N={N} nsims={nsims} ntheta={nth} {name:28s}: wall {1e6 * best:5.1f} us / iteration of 30; line search {ls:5.1f}, converged at start {cv:5.1f}, "
              f"steady median {1e6 * float(np.median(hist[5:, -1])):5.1f}; a {n3}-iteration call {1e6 * short:6.1f} us", flush=True)
lib.muse_debug_flags(prob._ctx, 0)
