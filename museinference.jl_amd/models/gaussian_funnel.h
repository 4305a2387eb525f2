/* gaussian_funnel.h -- the built-in funnel (MUSE_MODEL_FUNNEL: z_i ~ N(0, e^theta_k), x_i ~ N(z_i, 1); the reference's example,
 * src/simple.jl:59-73) written as a user-supplied model (include/muse_model.h), operation for operation as csrc/models.hpp's
 * FunnelModel: a template for a model of one's own, and the proof that the user-model seam costs nothing --
 * tests/test_user_model.py checks that this header and the built-in model give the same bits, bench.py (extra.user_model)
 * that they run at the same speed. */
#include "muse_model.h"
#define MUSE_MODEL_NAME "gaussian_funnel"
#define MUSE_MODEL_SECOND 1

MUSE_MODEL_FN void muse_model_sample(double sd, double n1, double n2, double* z, double* x, long i) {
    (void)i;
    *z = sd * n1;
    *x = *z + n2;
}
MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc, long i) {
    (void)i;
    const double r = x - z, t = iv * z;
    *acc = fma(t, z, fma(r, r, *acc));
    return t - r;
}
MUSE_MODEL_FN double muse_model_score_term(double x, double z, long i) {
    (void)i;
    (void)x;
    return z * z;
}

/* second derivatives (the implicit-differentiation get_H!): o = 1/2 [(x - z)^2 + iv z^2], B = z^2, x = sd n1 + n2 */
MUSE_MODEL_FN void muse_model_second(double iv, double x, double z, double* ozz, double* ozx, double* bz, double* bx, long i) {
    (void)i;
    (void)x;
    *ozz = 1.0 + iv;
    *ozx = -1.0;
    *bz = 2.0 * z;
    *bx = 0.0;
}
MUSE_MODEL_FN double muse_model_dx_dsd(double sd, double n1, double n2, long i) {
    (void)i;
    (void)sd;
    (void)n2;
    return n1;
}
