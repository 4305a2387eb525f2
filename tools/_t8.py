import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import museinference_jl_amd as M
def run(nth, theta, N=10000, nsims=512, z0=0, placement=None):
    prob = M.HipMuseProblem(None, model="funnel", ntheta=nth, N=N)
    prob.set_normals_cache(False); prob.set_timing(True)
    if placement is not None: prob.set_placement(placement)
    ks = []
    for _ in range(6):
        g, info = prob.map_and_score_batch(1, 0, nsims, theta, atol=1e-2, z0_mode=z0)
        ks.append(prob.last_kernel_ms() * 1e3)
    prob.close()
    return round(min(ks), 1), float(info["iterations"].mean())
for nth in (1, 4, 5, 8):
    th = np.linspace(-0.5, 1.0, nth) if nth > 1 else [0.25]
    print(nth, "equal thetas", run(nth, [1.0] * nth), "spread", run(nth, th), "streaming N=1e4", run(nth, th, placement=0), "N=30000", run(nth, th, N=30000), "N=4096", run(nth, th, N=4096), flush=True)
# the loop kernels at small N: device loop against host loop
for N in (2048, 4096, 10000, 30000):
    x, _ = M.HipMuseProblem(None, model="funnel", ntheta=1, N=N).sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0])
    prob = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
    out = []
    for dev in (False, True, False, True):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            n = prob.run_muse(0, [1.0], nsims=512, maxsteps=30, theta_rtol=1e-12, atol=1e-2, alpha=0.7, device_loop=dev)[0]
            best = min(best, (time.perf_counter() - t0) / n)
        out.append(("dev" if dev else "host", round(best * 1e6, 1)))
    print("loop N", N, out, flush=True)
    prob.close()
