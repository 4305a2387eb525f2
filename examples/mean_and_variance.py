"""A model with a LOCATION parameter: two populations of latent variables with unknown means and unknown log-variances, seen through
unit noise,

    z_i ~ Normal(mu_k, exp(tau_k / 2)),      x_i ~ Normal(z_i, 1),      theta = (mu_0, mu_1, tau_0, tau_1)

-- a header of the two-parameter family (include/muse_model.h, MUSE_MODEL_PAIR; the shipped museinference.jl_amd/models/
normal_mean_var.h): every block of elements has TWO parameters, and the header states how they enter the draw, the objective and
the score.  With the reference this is a SimpleMuseProblem whose closures capture nothing (src/simple.jl:79-95).  The latent field
integrates out here (x_i ~ Normal(mu_k, sqrt(1 + e^tau_k))), so the answer can be checked against the exact posterior.

    python examples/mean_and_variance.py     (needs an MI355X; there is no CPU path)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import museinference_jl_amd as M

model = M.ElementwiseModel.packaged("normal_mean_var")
N, K = 20000, 2
truth = np.array([0.5, -1.0, 0.0, 1.5])     # (mu_0, mu_1, tau_0, tau_1)

sim = M.HipMuseProblem(None, model=model, ntheta=2 * K, N=N)
print("consistency of the hand-written gradient and score:", M.check_model_consistency(sim, truth))
x, _ = sim.sample_x_z(M.SimRng(101, M.DATA_SIM), truth)
sim.close()

prob = M.HipMuseProblem(x, model=model, ntheta=2 * K, prior=M.GaussianPrior(0.0, 3.0))
result = M.muse(prob, np.zeros(2 * K), nsims=200, rng=0, grad_z_logLike_atol=1e-6, theta_rtol=1e-3, get_covariance=True)
sigma = np.sqrt(np.diag(result.Sigma))
# the exact maximum-likelihood values of the marginal model, per block: the sample mean, and log(sample variance - 1)
k = (np.arange(N) * K) // N
exact = np.array([x[k == b].mean() for b in range(K)] + [np.log(x[k == b].var() - 1.0) for b in range(K)])
names = ["mu_0", "mu_1", "tau_0", "tau_1"]
for j in range(2 * K):
    print(f"theta[{names[j]}] = {result.theta[j]:+.4f} +- {sigma[j]:.4f}    (truth {truth[j]:+.1f}: {abs(result.theta[j] - truth[j]) / sigma[j]:.2f} sigma;"
          f" exact marginal MLE {exact[j]:+.4f})")
print(f"{len(result.history)} iterations, {result.time * 1e3:.1f} ms")
assert np.all(np.abs(result.theta - truth) / sigma < 4.0)      # the reference's own acceptance criterion (test/runtests.jl:31)
assert np.all(np.abs(result.theta - exact) / sigma < 0.5)      # and the exact marginal answer, to the Monte-Carlo error of 200 sims
prob.close()
