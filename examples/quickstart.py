"""The reference's documentation example (docs/src/index.md:44-110, 150-184: the 2048-dimensional noisy funnel) on the MI355X
engine: theta ~ Normal(0, 3), z_i ~ Normal(0, exp(theta/2)), x_i ~ Normal(z_i, 1); data at theta = 0; MUSE estimate with its
covariance -- next to the exact marginal posterior, which this Gaussian problem happens to have in closed form.

    python examples/quickstart.py            (needs an MI355X; there is no CPU path)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import brentq

import museinference_jl_amd as M

N, truth, prior = 2048, 0.0, M.GaussianPrior(0.0, 3.0)

# "observations": one draw of the model at the true theta                      (docs/src/index.md:65-70)
sim = M.HipMuseProblem(None, model="funnel", ntheta=1, N=N)
x, _ = sim.sample_x_z(M.SimRng(1234, M.DATA_SIM), [truth])
sim.close()

# the problem = data + model + prior (SimpleMuseProblem(x, sample_x_z, logLike, logPrior), docs/src/index.md:153-172)
prob = M.HipMuseProblem(x, model="funnel", ntheta=1, prior=prior)

# muse(prob, theta0; nsims, get_covariance)                                    (docs/src/index.md:86-93, 174-180)
result = M.muse(prob, [0.0], nsims=500, rng=0, get_covariance=True)
print(f"MUSE:   theta = {result.theta[0]:+.4f} +- {np.sqrt(result.Sigma[0, 0]):.4f}   "
      f"({len(result.history)} iterations, {result.time * 1e3:.1f} ms)")

# the exact marginal posterior of this problem: x_i ~ Normal(0, 1 + exp(theta))
s2 = float(np.sum(x ** 2))
mode = brentq(lambda t: 0.5 * np.exp(t) / (1 + np.exp(t)) ** 2 * (s2 - N * (1 + np.exp(t))) - t / 9.0, -8, 8)
w = np.exp(mode) / (1 + np.exp(mode))
sigma = 1 / np.sqrt(0.5 * N * w * w + 1 / 9.0)
print(f"exact:  theta = {mode:+.4f} +- {sigma:.4f}")
assert abs(result.theta[0] - mode) < 4 * sigma / np.sqrt(500) and abs(np.sqrt(result.Sigma[0, 0]) / sigma - 1) < 0.2
prob.close()
