"""Wall time per outer iteration of the native muse! loops (muse_run: algebra on the host; muse_run_device: step kernel on
the GPU, no host round trip between maps) at configs[1]'s shape: python tools/runloop_bench.py [N] [nsims] [ntheta]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import museinference_jl_amd as M
argv = [a for a in sys.argv[1:] if a not in ("host", "dev", "once")]
HOST_ONLY, DEV_ONLY, ONCE = "host" in sys.argv[1:], "dev" in sys.argv[1:], "once" in sys.argv[1:]   # (profiling passes: one loop only / one run of it)
N, nsims, nth = (int(argv[0]), int(argv[1]), int(argv[2])) if len(argv) > 2 else (10000, 512, 1)
xdata, _ = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0] * nth)
prob = M.HipMuseProblem(xdata, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
if os.environ.get("MUSE_LOOP_DEBUG"):   # muse_debug_flags: 4 no prefetch, 8 the old element order, 32 no speculation
    M.load_library().muse_debug_flags(prob._ctx, int(os.environ["MUSE_LOOP_DEBUG"]))
for dev in ((False,) if HOST_ONLY else (True,) if DEV_ONLY else (False, True, False, True)):
    best = 1e9
    for rep in range(1 if ONCE else 5):
        t0 = time.perf_counter()
        n, theta, hist, gs, info = prob.run_muse(0, [1.0] * nth, nsims=nsims, maxsteps=30, theta_rtol=1e-12, atol=1e-2, alpha=0.7,
                                                 device_loop=dev)
        best = min(best, (time.perf_counter() - t0) / n)
    # the same loop ten times as long: the difference is the iterations' own time, without what a call costs once
    t0 = time.perf_counter()
    n2 = prob.run_muse(0, [1.0] * nth, nsims=nsims, maxsteps=300, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=dev)[0]
    slope = (time.perf_counter() - t0 - best * n) / max(1, n2 - n)
    print(f"N={N} nsims={nsims} ntheta={nth} device_loop={dev}: {1e6 * best:.1f} us per outer iteration ({n} iterations), "
          f"device-side iteration time {1e6 * float(np.median(hist[5:, -1])):.1f} us, marginal {1e6 * slope:.1f} us per iteration "
          f"(from a {n2}-iteration run)", flush=True)
