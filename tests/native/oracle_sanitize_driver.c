/* The CPU oracle (oracle/muse_oracle.c) under AddressSanitizer / UndefinedBehaviorSanitizer (tests/test_sanitizers.py):
 * every entry point the parity tests call, on small awkward shapes (odd N, N = 1, N not divisible by ntheta, several
 * threads), with heap buffers of exactly the documented sizes so that an out-of-bounds access is caught.  Exit 0 and
 * silence = clean.  (Test infrastructure checking test infrastructure: nothing here is the product.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t iterations, f_calls, status, hist_words; double f_min, gnorm; } mo_info;
void mo_sample_x_z(int model, int64_t N, int ntheta, uint64_t seed, uint64_t sim, const double* theta, double* x, double* z);
double mo_logLike_and_grad_z(int model, int64_t N, int ntheta, const double* x, const double* z, const double* theta, double* g);
void mo_grad_theta(int model, int64_t N, int ntheta, const double* x, const double* z, const double* theta, double* out);
int mo_zhat_at_theta(int model, int64_t N, int ntheta, const double* x, const double* z0, const double* theta, double atol,
                     double* zout, mo_info* info);
int mo_map_and_score_batch(int model, int64_t N, int ntheta, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                           const double* x_data, const double* theta, double atol, int z0_mode, double* zhat, double* g_out,
                           mo_info* info, int nthreads);
int mo_fd_jacobian(int model, int64_t N, int ntheta, uint64_t seed, int64_t sim, const double* theta0, const double* step,
                   double atol, const double* zfid, double* H_out);
int mo_implicit_H(int model, int64_t N, int ntheta, uint64_t seed, int64_t sim, const double* theta0, double atol, int cg_maxiter,
                  double* H_out, int32_t* cg_iters_out);

static double* vec(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

int main(void) {
    const int models[3] = {0, 1, 2};                 /* funnel, noise, smooth */
    const int64_t Ns[5] = {1, 7, 37, 130, 513};
    int bad = 0;
    for (int mi = 0; mi < 3; ++mi)
        for (int ni = 0; ni < 5; ++ni)
            for (int nt = 1; nt <= 3; nt += 2) {
                const int model = models[mi];
                const int64_t N = Ns[ni];
                if (model == 1 && nt > 1) continue;          /* the noise model has one theta */
                if (nt > N) continue;
                double theta[3] = {0.3, -0.4, 0.9}, step[3] = {0.05, 0.04, 0.03};
                double *x = vec(N), *z = vec(N), *g = vec(N), *zh = vec(N), *sc = vec(nt), *H = vec(nt * nt), *zfid = vec(N);
                mo_info info;
                mo_sample_x_z(model, N, nt, 5, 11, theta, x, z);
                const double f = mo_logLike_and_grad_z(model, N, nt, x, z, theta, g);
                mo_grad_theta(model, N, nt, x, z, theta, sc);
                mo_zhat_at_theta(model, N, nt, x, zfid, theta, 1e-6, zh, &info);
                if (!(info.iterations >= 0 && isfinite(f))) bad++;
                mo_fd_jacobian(model, N, nt, 5, 3, theta, step, 1e-4, zh, H);
                int32_t it[3];
                mo_implicit_H(model, N, nt, 5, 3, theta, 1e-1, 50, H, it);
                const int64_t nb = 6;
                double *zb = vec((nb + 1) * N), *gb = vec((nb + 1) * nt);
                mo_info* ib = (mo_info*)calloc(nb + 1, sizeof(mo_info));
                for (int z0_mode = 0; z0_mode <= 2; ++z0_mode)
                    mo_map_and_score_batch(model, N, nt, 5, 2, 2 + nb, 1, x, theta, 1e-3, z0_mode, zb, gb, ib, 3);
                for (int64_t k = 0; k <= nb; ++k)
                    if (!isfinite(gb[k * nt])) bad++;
                free(x); free(z); free(g); free(zh); free(sc); free(H); free(zfid); free(zb); free(gb); free(ib);
            }
    if (bad) { fprintf(stderr, "%d non-finite results\n", bad); return 1; }
    printf("oracle sanitize driver ok\n");
    return 0;
}
