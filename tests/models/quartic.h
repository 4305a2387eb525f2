/* A strictly convex NON-quadratic MAP objective for tests/test_gpu_linesearch.py (every built-in model is Gaussian, i.e.
 * quadratic in z, and on a quadratic HagerZhang ends with its first secant step): the noise-scale member of the family of
 * include/muse_model.h with a quartic term in the factor that does not depend on theta,
 *     F = sum_i 1/2 z_i^2 + 1/4 z_i^4 + 1/2 e^-theta (x_i - z_i)^2 (+ N theta / 2):   A = z^2 + z^4 / 2,  B = (x - z)^2
 * (lambda_min(Hessian) >= 1).  The draw is the noise model's (the test supplies its own x and starting points). */
#include "muse_model.h"
#define MUSE_MODEL_NAME "quartic"

MUSE_MODEL_FN void muse_model_sample(double sd, double n1, double n2, double* z, double* x, long i) {
    (void)i;
    *z = n1;
    *x = n1 + sd * n2;
}
MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc, long i) {
    (void)i;
    const double r = x - z, t = iv * r, z2 = z * z;
    *acc = fma(0.5, z2 * z2, fma(z, z, fma(t, r, *acc)));
    return fma(z2, z, z - t);
}
MUSE_MODEL_FN double muse_model_score_term(double x, double z, long i) {
    (void)i;
    const double r = x - z;
    return r * r;
}
