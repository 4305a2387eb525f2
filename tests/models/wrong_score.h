/* A header whose score term is NOT the B of its objective (z^2 in the objective, 1.1 z^2 in the score): what
 * museinference_jl_amd.check_model_consistency is there to catch (tests/test_user_model.py). */
#include "muse_model.h"
#define MUSE_MODEL_NAME "wrong_score"

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
    return 1.1 * (z * z);
}
