"""Soak test (development aid): contexts created and destroyed in a loop with every round-3 feature touched -- lanes,
multi-map launches, both native loops, the raw finite-difference seam, a communicator of each transport -- watching
device memory for leaks; then a minute of the pipelined two-lane step loop.  python tools/soak.py [cycles] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import museinference_jl_amd as M
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
free0 = None
for k in range(cycles):
    model, N, nth = [("funnel", 10000, 1), ("funnel", 3000, 4), ("noise", 70000, 1), ("smooth", 5000, 2)][k % 4]
    x = np.random.default_rng(k).standard_normal(N)
    p = M.HipMuseProblem(x, model=model, ntheta=nth, prior=M.GaussianPrior(0.0, 3.0))
    p.set_concurrency(1 + k % 4)
    th = np.full(nth, 0.3)
    for a in range(6):
        p.map_and_score_batch_async(k, 0, 20, th, result_area=a % 4)
        if a >= 3:
            p.batch_wait(20, (a - 3) % 4)
    for a in range(3, 6):
        p.batch_wait(20, a % 4)
    tot = p.map_and_score_multi_async(k, 0, 9, np.tile(th, (3, 1)), include_data=True, result_area=1)
    p.batch_wait(tot, 1)
    p.run_muse(k, th, nsims=8, maxsteps=3, theta_rtol=0.0, atol=1e-2, alpha=0.5, device_loop=bool(k % 2))
    p.fd_values_columns(k, 0, 0, 2 * nth, th, np.full((nth, 2), 0.01))
    p.comm_init(1, 0, M.HipMuseProblem.comm_unique_id("shm" if k % 2 else "rccl"))
    n = p.map_and_score_batch_gather_async(k, 0, 12, th, 16)
    p.batch_wait_gathered(n, 16)
    p.comm_destroy()
    p.close()
    free, total = torch.cuda.mem_get_info()
    if k == 8:
        free0 = free
    if k % 20 == 0:
        print(f"cycle {k}: device memory free {free / 2**20:.0f} MiB", flush=True)
if free0 is not None:
    leak = (free0 - free) / 2**20
    print(f"free memory after cycle 8: {free0 / 2**20:.0f} MiB, after cycle {cycles - 1}: {free / 2**20:.0f} MiB (difference {leak:.1f} MiB)")
    assert leak < 64, "device memory is leaking"
p = M.HipMuseProblem(None, model="funnel", ntheta=1, N=10000)
p.set_concurrency(2)
t0, steps = time.time(), 0
pend = []
while time.time() - t0 < seconds:
    pend.append(p.map_and_score_batch_async(0, 0, 512, [1.0], result_area=steps % 4))
    if len(pend) > 3:
        g, info = p.batch_wait(pend.pop(0), (steps - 3) % 4)
        assert np.all(info["status"] == 0)
    steps += 1
while pend:
    p.batch_wait(pend.pop(0), (steps - len(pend) - 1) % 4)
p.synchronize()
print(f"{steps} pipelined two-lane steps in {time.time() - t0:.1f} s: {1e6 * (time.time() - t0) / steps:.1f} us per step, all converged "
      f"(the loop repeats ONE step: from its third pass on the normals come from the cross-call cache, not from the generator -- "
      f"not bench.py's step, which draws its simulations)")
