"""Per-element constants in a user-supplied model -- what a closure of the reference's SimpleMuseProblem would capture: the
amplitude A = exp(theta) of a field with a KNOWN power spectrum P_i observed in unit white noise,

    z_i ~ Normal(0, exp(theta) P_i),      x_i ~ Normal(z_i, 1),

with the spectrum compiled into the model's library as a table (ElementwiseModel.from_source(..., constants={"P": P}): the
model's three functions read it as P(i)).  This one is jointly Gaussian, so the exact marginal posterior is there to compare with.

    python examples/spectrum.py              (needs an MI355X; there is no CPU path)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import brentq

import museinference_jl_amd as M

SOURCE = r'''
#define MUSE_MODEL_NAME "known_spectrum"
MUSE_MODEL_FN void muse_model_sample(double sd, double n1, double n2, double* z, double* x, long i) {
    *z = (sd * sqrt(P(i))) * n1;
    *x = *z + n2;
}
MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc, long i) {
    const double r = x - z, t = (iv / P(i)) * z;        /* -logLike = 1/2 sum [(x - z)^2 + e^-theta z^2 / P_i] + N theta / 2 */
    *acc = fma(t, z, fma(r, r, *acc));
    return t - r;
}
MUSE_MODEL_FN double muse_model_score_term(double x, double z, long i) { (void)x; return (z * z) / P(i); }
'''

N, truth = 10000, 0.4
P = 30.0 / (1.0 + np.arange(N) % 250) ** 1.7 + 0.02            # signal-to-noise from 30 down to 0.02, repeated
model = M.ElementwiseModel.from_source("known_spectrum", SOURCE, constants={"P": P})

sim = M.HipMuseProblem(None, model=model, ntheta=1, N=N)
x, _ = sim.sample_x_z(M.SimRng(11, M.DATA_SIM), [truth])
sim.close()

prob = M.HipMuseProblem(x, model=model, ntheta=1, prior=M.GaussianPrior(0.0, 3.0))
result = M.muse(prob, [0.0], nsims=400, rng=0, grad_z_logLike_atol=1e-6, theta_rtol=1e-3, get_covariance=True)
print(f"MUSE:   theta = {result.theta[0]:+.4f} +- {np.sqrt(result.Sigma[0, 0]):.4f}   ({len(result.history)} iterations, {result.time * 1e3:.1f} ms)")

# exact: x_i ~ Normal(0, 1 + exp(theta) P_i)
score = lambda t: 0.5 * np.sum(np.exp(t) * P * (x ** 2 - (1 + np.exp(t) * P)) / (1 + np.exp(t) * P) ** 2) - t / 9.0
mode = brentq(score, -8, 8)
w = np.exp(mode) * P / (1 + np.exp(mode) * P)
sigma = 1 / np.sqrt(0.5 * np.sum(w ** 2) + 1 / 9.0)
print(f"exact:  theta = {mode:+.4f} +- {sigma:.4f}      (truth {truth:+.1f})")
assert abs(result.theta[0] - mode) < 4 * sigma / np.sqrt(400) and abs(np.sqrt(result.Sigma[0, 0]) / sigma - 1) < 0.25
prob.close()
