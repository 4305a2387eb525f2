/* normal_mean_var.h -- a model of the TWO-PARAMETER family (include/muse_model.h, MUSE_MODEL_PAIR): every block has a location
 * AND a scale parameter acting on the same elements,
 *
 *     z_i ~ N(mu_k, e^tau_k),   x_i ~ N(z_i, 1)          k = the element's block, theta = (mu_0 .. mu_{K-1}, tau_0 .. tau_{K-1})
 *
 * -logLike = 1/2 sum_i [ (x_i - z_i)^2 + e^-tau_k (z_i - mu_k)^2 ] + 1/2 sum_k n_k tau_k.  The latent field integrates out in
 * closed form (x_i ~ N(mu_k, 1 + e^tau_k) independently), so MUSE's estimate can be compared with the exact marginal posterior
 * (tests/test_user_model.py).  As a SimpleMuseProblem of the reference (src/simple.jl:79-95) this is
 *     sample_x_z = (rng, th) -> (z = th.mu .+ exp.(th.tau ./ 2) .* randn(rng, N); x = z .+ randn(rng, N); (; x, z))
 *     logLike    = (x, z, th) -> -(sum((x .- z).^2) + sum(exp.(-th.tau) .* (z .- th.mu).^2) + N * th.tau) / 2
 * with one block. */
#define MUSE_MODEL_PAIR 1
#include "muse_model.h"
#define MUSE_MODEL_NAME "normal_mean_var"

/* c = { mu, sd = e^(tau/2), iv = e^-tau, (unused) };  the block's constant per element: C = tau */
MUSE_MODEL_FN double muse_model_coefs(double mu, double tau, double* c) {
    c[0] = mu;
    c[1] = muse_model_exp(0.5 * tau);
    c[2] = muse_model_exp(-tau);
    c[3] = 0.0;
    return tau;
}
MUSE_MODEL_FN void muse_model_sample(const double* c, double n1, double n2, double* z, double* x, long i) {
    (void)i;
    *z = fma(c[1], n1, c[0]);
    *x = *z + n2;
}
/* d(1/2 o)/dz = iv (z - mu) - (x - z);  o = (x - z)^2 + iv (z - mu)^2 */
MUSE_MODEL_FN double muse_model_grad(const double* c, double x, double z, double* acc, long i) {
    (void)i;
    const double r = x - z, d = z - c[0], t = c[2] * d;
    *acc = fma(t, d, fma(r, r, *acc));
    return t - r;
}
MUSE_MODEL_FN void muse_model_score_terms(const double* c, double x, double z, double* t0, double* t1, long i) {
    (void)x; (void)i;
    const double d = z - c[0];
    *t0 = d;
    *t1 = d * d;
}
/* d logLike / d mu = iv sum (z - mu);   d logLike / d tau = 1/2 (iv sum (z - mu)^2 - n) */
MUSE_MODEL_FN void muse_model_score(const double* c, double S0, double S1, double n, double* ga, double* gb) {
    *ga = c[2] * S0;
    *gb = 0.5 * (c[2] * S1 - n);
}
