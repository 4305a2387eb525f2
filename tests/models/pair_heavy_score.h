/* TEST MODEL (tests/test_symbolic_model.py): models/normal_mean_var.h with the factor e^-tau applied to every ELEMENT's score terms
 * instead of to the block's sums -- the same model, and a correct header.  In the LDS-resident solver kernels the extra
 * coefficient read in the last pass costs 500 more spilled registers (loop kernel of three or four blocks: 634 spilled VGPRs, 420
 * bytes of scratch per lane, against 126 / 120): the header that showed that a user's model can push a loop kernel past the product's
 * bound on scratch -- muse_run_device then runs the host loop (muse_engine.cpp, loop_usable). */
#define MUSE_MODEL_PAIR 1
#include "muse_model.h"
#define MUSE_MODEL_NAME "pair_heavy_score"

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
    *t0 = c[2] * d;
    *t1 = 0.5 * (c[2] * (d * d));
}
/* d logLike / d mu = iv sum (z - mu);   d logLike / d tau = 1/2 (iv sum (z - mu)^2 - n) */
MUSE_MODEL_FN void muse_model_score(const double* c, double S0, double S1, double n, double* ga, double* gb) {
    (void)c;
    *ga = S0;
    *gb = S1 - 0.5 * n;
}
