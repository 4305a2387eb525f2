"""The two native muse! loops side by side over theta components and placements: per-iteration time (the loop's own clock, median from
iteration 6 on; wall per iteration of a 30-iteration call) of muse_run (one launch per iteration, step on the host) and of the loop
kernel (MUSE_DEBUG_LOOP_ANY_NTHETA=1: whatever ntheta) -- what muse_run_device's routing is decided from.
    MUSE_DEBUG_LOOP_ANY_NTHETA=1 python tools/loop_vs_host.py [nsims]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import museinference_jl_amd as M

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for N in (10000, 4096, 512):
    for nth in (1, 2, 4, 8):
        xdata, _ = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0] * nth)
        prob = M.HipMuseProblem(xdata, model="funnel", ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
        out = {}
        for dev in (False, True, False, True):
            best, steady = 1e9, 0.0
            for _ in range(3):
                t0 = time.perf_counter()
                n, theta, hist, gs, info = prob.run_muse(0, [1.0] * nth, nsims=S, maxsteps=30, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=dev)
                best = min(best, (time.perf_counter() - t0) / n)
                steady = float(np.median(hist[5:, -1]))
            k = "device" if dev else "host"
            out[k] = (min(out.get(k, (1e9, 1e9))[0], 1e6 * best), min(out.get(k, (1e9, 1e9))[1], 1e6 * steady))
        print(f"N={N:6d} ntheta={nth} nsims={S}: host loop {out['host'][0]:6.1f} wall / {out['host'][1]:6.1f} steady us   "
              f"loop kernel {out['device'][0]:6.1f} wall / {out['device'][1]:6.1f} steady us", flush=True)
        prob.close()
