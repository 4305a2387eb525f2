/* cubic.h -- example of a user-supplied elementwise model (include/muse_model.h): a funnel seen through a NON-LINEAR
 * observation function,
 *     z_i ~ N(0, e^theta_k),   x_i ~ N(h(z_i), 1),   h(z) = z + z^3 / 10        (h' = 1 + 3 z^2 / 10 > 0)
 * so that  -logLike = 1/2 sum_i [ (x_i - h(z_i))^2 + e^-theta_k z_i^2 ] + 1/2 sum_k n_k theta_k :
 * A = (x - h(z))^2, B = z^2.  The MAP objective is not quadratic in z (the line search does real work, unlike on the
 * Gaussian models), the posterior is not Gaussian, and MUSE's estimate of theta is still unbiased. */
#include "muse_model.h"
#define MUSE_MODEL_NAME "cubic"
#define MUSE_MODEL_SECOND 1   /* second derivatives below: the implicit-differentiation get_H! accepts the model */

MUSE_MODEL_FN void muse_model_sample(double sd, double n1, double n2, double* z, double* x, long i) {
    (void)i;
    const double zi = sd * n1;
    *z = zi;
    *x = fma(0.1 * (zi * zi), zi, zi) + n2;
}
MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc, long i) {
    (void)i;
    const double z2 = z * z;
    const double r = x - fma(0.1 * z2, z, z);   /* x - h(z) */
    const double t = iv * z;
    *acc = fma(t, z, fma(r, r, *acc));
    return t - r * fma(0.3, z2, 1.0);           /* iv z - (x - h) h'(z) */
}
MUSE_MODEL_FN double muse_model_score_term(double x, double z, long i) {
    (void)i;
    (void)x;
    return z * z;
}

/* With o = 1/2 [(x - h)^2 + iv z^2]:  do/dz = iv z - (x - h) h',  d2o/dz2 = iv + h'^2 - (x - h) h'',  d2o/dzdx = -h'  (h'' = 0.6 z);
 * B = z^2: dB/dz = 2 z, dB/dx = 0;  x = h(sd n1) + n2: dx/dsd = h'(sd n1) n1. */
MUSE_MODEL_FN void muse_model_second(double iv, double x, double z, double* ozz, double* ozx, double* bz, double* bx, long i) {
    (void)i;
    const double z2 = z * z;
    const double r = x - fma(0.1 * z2, z, z);
    const double hp = fma(0.3, z2, 1.0);
    *ozz = fma(hp, hp, iv) - r * (0.6 * z);
    *ozx = -hp;
    *bz = 2.0 * z;
    *bx = 0.0;
}
MUSE_MODEL_FN double muse_model_dx_dsd(double sd, double n1, double n2, long i) {
    (void)i;
    (void)n2;
    const double zi = sd * n1;
    return fma(0.3, zi * zi, 1.0) * n1;
}
