"""A user-supplied model (the closures of SimpleMuseProblem, src/simple.jl:79-95, as a C header: include/muse_model.h): two
blocks of latent variables with unknown log-variances seen through a saturating, non-linear detector,

    z_i ~ Normal(0, exp(theta_k / 2)),      x_i ~ Normal(h(z_i), 1),      h(z) = z / sqrt(1 + z^2 / 16)

so the posterior of z is not Gaussian and the marginal likelihood has no closed form -- the case MUSE is for.  The header is
compiled (hipcc, ~1 min, once) into an engine library of its own; everything else is the same API.

    python examples/user_model.py            (needs an MI355X; there is no CPU path)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import museinference_jl_amd as M

SOURCE = r'''
#include "muse_model.h"
#define MUSE_MODEL_NAME "saturating"
/* -logLike = 1/2 sum_i [ (x_i - h(z_i))^2 + exp(-theta_k) z_i^2 ] + 1/2 sum_k n_k theta_k :   A = (x - h)^2,  B = z^2 */
MUSE_MODEL_FN void muse_model_sample(double sd, double n1, double n2, double* z, double* x, long i) {
    (void)i;
    const double zi = sd * n1;
    *z = zi;
    *x = zi / sqrt(fma(0.0625 * zi, zi, 1.0)) + n2;
}
MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc, long i) {
    (void)i;
    const double q = fma(0.0625 * z, z, 1.0);       /* 1 + z^2/16 */
    const double rq = 1.0 / sqrt(q);
    const double r = x - z * rq;                    /* x - h(z)   */
    const double t = iv * z;
    *acc = fma(t, z, fma(r, r, *acc));
    return t - r * (rq / q);                        /* iv z - (x - h) h'(z),  h' = (1 + z^2/16)^(-3/2) */
}
MUSE_MODEL_FN double muse_model_score_term(double x, double z, long i) { (void)x; (void)i; return z * z; }
'''

model = M.ElementwiseModel.from_source("saturating", SOURCE)
N, truth = 20000, np.array([1.0, 2.0])

sim = M.HipMuseProblem(None, model=model, ntheta=2, N=N)
print("consistency of the hand-written gradient and score:", M.check_model_consistency(sim, truth))
x, _ = sim.sample_x_z(M.SimRng(101, M.DATA_SIM), truth)
sim.close()

prob = M.HipMuseProblem(x, model=model, ntheta=2, prior=M.GaussianPrior(0.0, 3.0))
result = M.muse(prob, [0.0, 0.0], nsims=200, rng=0, grad_z_logLike_atol=1e-4, theta_rtol=1e-2, get_covariance=True)
sigma = np.sqrt(np.diag(result.Sigma))
for k in range(2):
    print(f"theta[{k}] = {result.theta[k]:+.4f} +- {sigma[k]:.4f}    (truth {truth[k]:+.1f}: {abs(result.theta[k] - truth[k]) / sigma[k]:.2f} sigma)")
print(f"{len(result.history)} iterations, {result.time * 1e3:.1f} ms")
assert np.all(np.abs(result.theta - truth) / sigma < 4.0)      # the reference's own acceptance criterion (test/runtests.jl:31)
prob.close()
