"""Diagnostic: wall time of a SYNCHRONOUS map call (launch -> results on the host) against the kernel's own duration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
p0 = M.HipMuseProblem(None, model="funnel", N=10000)
x, _ = p0.sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0])
prob = M.HipMuseProblem(x, model="funnel")
prob.map_and_score_batch(0, 0, 512, [1.0], include_data=True)
for timing in (True, False):
    prob.set_timing(timing)
    for _ in range(20): prob.map_and_score_batch(0, 0, 512, [1.0], include_data=True, z0_mode=M.Z0_WARM)
    t0 = time.perf_counter()
    for _ in range(200): prob.map_and_score_batch(0, 0, 512, [1.0], include_data=True, z0_mode=M.Z0_WARM)
    dt = (time.perf_counter() - t0) / 200
    print(f"timing events {timing}: {dt*1e6:.1f} us per synchronous warm map (kernel {prob.last_kernel_ms()*1e3 if timing else float('nan'):.1f} us)")
