/* The built-in funnel (MUSE_MODEL_FUNNEL: z ~ N(0, e^theta), x ~ N(z, 1)) written as a user's model, operation for
 * operation as models.hpp's FunnelModel: tests/test_user_model.py checks that the two give the same bits. */
#include "muse_model.h"
#define MUSE_MODEL_NAME "funnel_as_user"

MUSE_MODEL_FN void muse_model_sample(double sd, double n1, double n2, double* z, double* x) {
    *z = sd * n1;
    *x = *z + n2;
}
MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc) {
    const double r = x - z, t = iv * z;
    *acc = fma(t, z, fma(r, r, *acc));
    return t - r;
}
MUSE_MODEL_FN double muse_model_score_term(double x, double z) {
    (void)x;
    return z * z;
}
