"""Diagnostic: per-launch kernel time inside muse_run (iteration 1 stores the normals, later ones load them)
and inside an FD batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, museinference_jl_amd as M
p0 = M.HipMuseProblem(None, model="funnel", N=10000)
x, _ = p0.sample_x_z(M.SimRng(0, M.DATA_SIM), [0.0])
prob = M.HipMuseProblem(x, model="funnel", prior=M.GaussianPrior(0.0, 3.0))
M.muse(prob, [1.0], rng=0, nsims=512, maxsteps=3)
prob.profile_begin(64)
M.muse(prob, [1.0], rng=0, nsims=512, maxsteps=8, theta_rtol=1e-12)
print("muse_run kernel us per iteration:", np.round(prob.profile_end() * 1e3, 1))
prob.fd_jacobian_batch(0, 0, 64, [1.0], [0.05])
prob.profile_begin(8)
prob.fd_jacobian_batch(0, 0, 64, [1.0], [0.05])
print("FD batch kernels us (prep, FD):", np.round(prob.profile_end() * 1e3, 1))
prob.profile_begin(8)
prob.fd_jacobian_batch(0, 0, 512, [1.0], [0.05])
print("FD batch 512 sims kernels us (prep, FD):", np.round(prob.profile_end() * 1e3, 1))
p4 = M.HipMuseProblem(x, model="funnel", ntheta=4)
p4.fd_jacobian_batch(0, 0, 64, [1.0] * 4, [0.05] * 4)
p4.profile_begin(8)
import time
t0 = time.perf_counter()
p4.fd_jacobian_batch(0, 0, 64, [1.0] * 4, [0.05] * 4)
dt = time.perf_counter() - t0
print("configs[3] per-GPU share (ntheta=4, 64 sims -> 512 FD problems): kernels us (prep, FD):", np.round(p4.profile_end() * 1e3, 1), "wall us", round(dt * 1e6, 1))
