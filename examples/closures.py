"""A model NO header holds, as closures: the reference's SimpleMuseProblem (src/simple.jl:4-12, 79-95) is TorchMuseProblem here.

    z ~ Normal(0, exp(theta / 2) I_24),      x ~ Normal(A z, I_40)        A: a dense 40 x 24 matrix

Every latent variable enters every observation, so this is not an elementwise model (include/muse_model.h) and cannot run on the HIP
solver; the general front-end runs it: sample_x_z and logLike are torch closures on the GPU, their derivatives come from autograd
(the reference: ForwardDiff / Zygote), the MAP from the interface's default solver (L-BFGS + HagerZhang on the device's tensors).
Marginally x ~ Normal(0, I + e^theta A A'): the exact posterior of theta is a one-dimensional integral, printed beside the estimate.

    python examples/closures.py          (tensors on the GPU when there is one)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import museinference_jl_amd as M

dev = "cuda" if torch.cuda.is_available() else "cpu"
n, m, truth = 24, 40, 0.4
A = torch.as_tensor(np.random.RandomState(2).randn(m, n) / np.sqrt(n), device=dev)


def sample_x_z(gen, theta):
    z = torch.exp(theta[0] / 2) * torch.randn(n, generator=gen, device=dev, dtype=torch.float64)
    return A @ z + torch.randn(m, generator=gen, device=dev, dtype=torch.float64), z


def logLike(x, z, theta):
    r = x - A @ z
    return -0.5 * (torch.sum(r * r) + torch.exp(-theta[0]) * torch.sum(z * z) + n * theta[0])


logPrior = lambda theta: -0.5 * torch.sum(theta ** 2) / 9.0
x, _ = M.TorchMuseProblem(None, sample_x_z, logLike, device=dev).sample_x_z(M.SimRng(100, M.DATA_SIM), [truth])
prob = M.TorchMuseProblem(x, sample_x_z, logLike, logPrior, device=dev)
result = M.muse(prob, [0.0], rng=0, nsims=50, maxsteps=30, theta_rtol=1e-3, grad_z_logLike_atol=1e-6, alpha=1.0, get_covariance=True)

lam, U = np.linalg.eigh((A @ A.T).cpu().numpy())
y2 = (U.T @ x.cpu().numpy()) ** 2
ts = np.linspace(-6, 6, 4001)
lp = np.array([-0.5 * np.sum(np.log1p(np.exp(t) * lam) + y2 / (1 + np.exp(t) * lam)) - 0.5 * t * t / 9.0 for t in ts])
w = np.exp(lp - lp.max())
w /= w.sum()
mean = float(np.sum(w * ts))
sd = float(np.sqrt(np.sum(w * (ts - mean) ** 2)))
sigma = float(np.sqrt(result.Sigma[0, 0]))
print(f"theta = {result.theta[0]:+.3f} +- {sigma:.3f}    (exact posterior: mode {ts[np.argmax(lp)]:+.3f}, mean {mean:+.3f} +- {sd:.3f}; truth {truth:+.1f})")
print(f"{len(result.history)} iterations on {dev}, {result.time:.2f} s")
assert abs(result.theta[0] - ts[np.argmax(lp)]) < 2.5 * sd and 0.3 < sigma / sd < 3.0
