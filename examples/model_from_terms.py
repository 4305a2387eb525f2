"""A model given by its TERMS only: the derivatives are generated.

The reference differentiates the closures of a SimpleMuseProblem by AD (src/simple.jl:84-85).  Device code is not differentiated
here -- a model header (include/muse_model.h) states its gradient, its score term and, for the implicit-differentiation get_H!
(src/muse.jl:335-405), its second derivatives.  `ElementwiseModel.from_expressions` writes that header from the two terms of the
one-parameter family,

    -logLike = 1/2 sum_i [ A(x_i, z_i) + e^-theta_k B(x_i, z_i) ] + 1/2 sum_k n_k theta_k,

and the draw, differentiating them symbolically (sympy): here a latent Gaussian field seen through a response that grows faster
than linearly, z_i ~ Normal(0, exp(theta_k / 2)), x_i ~ Normal(z_i sqrt(1 + z_i^2), 1).

    python examples/model_from_terms.py     (needs an MI355X; there is no CPU path)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import museinference_jl_amd as M

TERMS = dict(A="(x - z*sqrt(1 + z**2))**2", B="z**2", z="sd*n1", x="z*sqrt(1 + z**2) + n2")
model = M.ElementwiseModel.from_expressions("growing_response", **TERMS)
print(open(model.header).read().split("*/")[0] + "*/\n  ... (" + model.header + ")")

N, truth = 20000, np.array([0.3, -0.6])
sim = M.HipMuseProblem(None, model=model, ntheta=2, N=N)
print("consistency of the generated derivatives (first and second):", M.check_model_consistency(sim, truth))
x, _ = sim.sample_x_z(M.SimRng(7, M.DATA_SIM), truth)
sim.close()

prob = M.HipMuseProblem(x, model=model, ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
result = M.muse(prob, np.zeros(2), nsims=200, rng=0, grad_z_logLike_atol=1e-6, theta_rtol=1e-3)
M.get_J_(result, prob, grad_z_logLike_atol=1e-6)
M.get_H_(result, prob, nsims=20, implicit_diff=True)          # through the generated second derivatives
H_implicit = result.H.copy()
result.Hs, result.H = [], None
M.get_H_(result, prob, nsims=20)                               # the finite-difference branch, same simulations
sigma = np.sqrt(np.diag(result.Sigma))
for j in range(2):
    print(f"theta[{j}] = {result.theta[j]:+.4f} +- {sigma[j]:.4f}    (truth {truth[j]:+.1f}: {abs(result.theta[j] - truth[j]) / sigma[j]:.2f} sigma)")
print("H by implicit differentiation:", np.diag(H_implicit), " by finite differences:", np.diag(result.H))
print(f"{len(result.history)} iterations, {result.time * 1e3:.1f} ms")
assert np.all(np.abs(result.theta - truth) / sigma < 4.0)      # the reference's own acceptance criterion (test/runtests.jl:31)
np.testing.assert_allclose(H_implicit, result.H, rtol=0.02, atol=0.02 * np.abs(result.H).max())
prob.close()
