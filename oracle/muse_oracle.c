/*
 * muse_oracle.c -- CPU restatement of the MUSE inner loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (museinference.jl_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" against reference *outputs*: the reference
 * (marius311/MuseInference.jl, Julia) cannot run in this image (no julia binary) and holds
 * no golden vectors or known-answer tests for this path (test/runtests.jl:31,56,81 only
 * assert |mu|/sigma < 2).  The oracle is instead pinned against
 *   (i)  closed-form known answers of every model (tests/test_oracle.py):
 *        funnel MAP  zhat = x/(1+e^-theta), score = 1/2 [e^-theta sum zhat^2 - N], ...
 *   (ii) scipy.optimize L-BFGS-B minimisers of the same objective,
 *   (iii) the Philox4x32-10 known-answer vectors of Salmon et al. (Random123 kat_vectors),
 *   (iv) the reference's own statistical criterion mu/sigma < 2 on its 512-dim funnel test,
 *   (v)  a published known answer of the un-vendored dependency: the work counters the Optim.jl documentation prints for
 *        LBFGS() on Rosenbrock's function from (0, 0) -- 24 iterations, 67 f/g evaluations -- which the restated
 *        LBFGS/HagerZhang below reproduces exactly (tests/test_oracle.py; quoted from memory, no network here).
 *   (vi) the exact marginal posterior of the (jointly Gaussian) models, for which MUSE is exact: muse() on this oracle returns
 *        the closed-form posterior mode to sigma/sqrt(nsims) and the exact Fisher information from get_J! and get_H!
 *        (tests/test_exact_marginal.py).
 *
 * What each function follows (paths relative to /root/reference):
 *   mo_sample_x_z           src/simple.jl:61-65, docs/src/index.md:156-160  (funnel closure)
 *   mo_logLike_and_grad_z   src/interface.jl:68-83, src/simple.jl:66-68,85
 *   mo_grad_theta           src/interface.jl:41-58, src/simple.jl:84
 *   mo_zhat_at_theta        src/interface.jl:162-171  (Optim.optimize(only_fg, z0, LBFGS(),
 *                           Options(g_tol=atol)) -- minimise -logLike)
 *   mo_map_and_score        src/muse.jl:169-176 (muse! map body), :508-525 (get_J! body)
 *   mo_fd_jacobian          src/muse.jl:426-442 + src/util.jl:9-27 (pjacobian, central_fdm(3,1))
 *   mo_implicit_H           src/muse.jl:335-405 (get_H! implicit-differentiation branch, CG solve)
 *
 * Third-party arithmetic that is NOT vendored in the reference (Project.toml compat only,
 * no Manifest.toml) and is restated here from the published algorithms:
 *   Optim.jl ("1.5" compat, Project.toml:45)  LBFGS(m=10, alphaguess=InitialStatic(alpha=1),
 *       linesearch=HagerZhang(), scaleinvH0=true); Options(x_abstol=x_reltol=f_abstol=
 *       f_reltol=0, g_abstol=atol, successive_f_tol=1, iterations=1000,
 *       allow_f_increases=true).
 *   LineSearches.jl HagerZhang (delta=.1, sigma=.9, alphamax=Inf, rho=5, epsilon=1e-6,
 *       gamma=.66, linesearchmax=50, psi3=.1): Hager & Zhang, "Algorithm 851: CG_DESCENT",
 *       ACM TOMS 32 (2006), stages B0-B3 (bracket), U0-U3 (update), S1-S4 (secant2).
 *   FiniteDifferences.jl central_fdm(3,1) (Project.toml:40): grid (-1,0,1), coefficients
 *       (-1/2,0,1/2), explicit step => no adaptation.
 * Random numbers: Julia's RNG streams cannot be reproduced without Julia; "same seeds"
 * means the build-defined counter-based stream below, shared (as an algorithm, not as
 * code) by this oracle and the HIP kernels:
 *   Philox4x32-10, key = (seed lo32, seed hi32), counter = (i lo32, i hi32, sim lo32,
 *   sim hi32); u1 = ((w0<<20 | w1>>12) + 0.5) 2^-52, u2 likewise from (w2,w3);
 *   Box-Muller r = sqrt(-2 log u1), n1 = r cos(2 pi u2), n2 = r sin(2 pi u2), with log and
 *   sin/cos(pi t) evaluated by the fixed fdlibm-style polynomial sequences written out
 *   below using only IEEE +,-,*,/,sqrt and explicit fma (no implicit contraction), so that both
 *   sides are bit-equal.
 * Model arithmetic: the reference's closures are plain Julia broadcasts; here the elementwise
 * objective/gradient and the L-BFGS vector updates are written with explicit fma in the places the
 * HIP kernels use it (x + alpha s, q - alpha dg, s + coef dx, sums of products), so that the two
 * sides differ only in reduction order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Per-thread workspace: the work vectors of a solve are carved from a thread-local arena that is
 * reserved once per thread and reused across sims (a malloc/mmap per vector per sim serialises in the
 * kernel when many OpenMP threads run sims -- it only matters for the timed CPU baseline). */
static __thread char* tl_arena = NULL;
static __thread size_t tl_cap = 0, tl_off = 0;
static void ws_reserve(int64_t N) {
    const size_t need = ((size_t)(40 + 2 * 10) * (size_t)N + 1024) * sizeof(double);
    if (need > tl_cap) {
        free(tl_arena);
        tl_arena = (char*)malloc(need);
        tl_cap = need;
        tl_off = 0;
    }
}
static double* ws_alloc(size_t n) {
    const size_t bytes = (n * sizeof(double) + 63) & ~(size_t)63;
    if (tl_off + bytes > tl_cap) return (double*)malloc(bytes); /* falls back (never freed by ws_release) */
    double* p = (double*)(tl_arena + tl_off);
    tl_off += bytes;
    return p;
}

#define MO_MODEL_FUNNEL 0 /* z_i ~ N(0, e^theta_k(i)), x_i ~ N(z_i, 1)          */
#define MO_MODEL_NOISE 1  /* z_i ~ N(0, 1),            x_i ~ N(z_i, e^theta)     */
#define MO_MODEL_SMOOTH 2 /* z as funnel, x = A z + n, A = periodic (1/4,1/2,1/4) */
/* A user-supplied elementwise model (include/muse_model.h; the engine compiles the header's three functions into a
 * library of its own): built with -DMO_USER_MODEL_HEADER="<header>" (make user, below in the Makefile) this checker
 * compiles the SAME text, so that the HIP path of a user's model has a CPU counterpart like the built-in ones:
 * -logLike = 1/2 sum_i [A(x_i,z_i) + e^-theta_k B(x_i,z_i)] + 1/2 sum_k n_k theta_k, score_k = 1/2 (e^-theta_k sum B - n_k). */
#define MO_MODEL_USER 3
#ifdef MO_USER_MODEL_HEADER
#define MUSE_MODEL_FN static inline
static double mo_exp(double x);              /* (defined below: the engine's fixed-sequence exp, restated) */
#define muse_model_exp(x) mo_exp(x)          /* what a header of the two-parameter family forms its coefficients with */
#include MO_USER_MODEL_HEADER
const char* mo_user_model_name(void) { return MUSE_MODEL_NAME; }
#ifdef MUSE_MODEL_NCONST /* run-time constants (include/muse_model.h: muse_const): this checker's copies */
const double* muse_host_consts[MUSE_MODEL_MAX_CONST] = {0, 0, 0, 0};
long muse_host_const_len[MUSE_MODEL_MAX_CONST] = {0, 0, 0, 0};
int mo_set_constants(int k, const double* values, int64_t count) {
    if (k < 0 || k >= MUSE_MODEL_NCONST || !values || count < 1) return -1;
    double* copy = (double*)malloc((size_t)count * sizeof(double));
    if (!copy) return -2;
    memcpy(copy, values, (size_t)count * sizeof(double));
    free((void*)muse_host_consts[k]);
    muse_host_consts[k] = copy;
    muse_host_const_len[k] = (long)count;
    return 0;
}
#else
int mo_set_constants(int k, const double* values, int64_t count) { (void)k; (void)values; (void)count; return -3; }
#endif
#ifdef MUSE_MODEL_N /* a model with per-element tables is built for one N */
#define MO_USER_CHECK_N(N) do { if ((N) != (int64_t)(MUSE_MODEL_N)) abort(); } while (0)
#else
#define MO_USER_CHECK_N(N) do { } while (0)
#endif
#else
const char* mo_user_model_name(void) { return 0; }
int mo_set_constants(int k, const double* values, int64_t count) { (void)k; (void)values; (void)count; return -3; }
#endif

#define MO_STATUS_G_CONVERGED 0
#define MO_STATUS_X_CONVERGED 1
#define MO_STATUS_F_CONVERGED 2
#define MO_STATUS_MAXITER 3
#define MO_STATUS_LINESEARCH_FAILED 4
#define MO_STATUS_NONFINITE 5

typedef struct {
    int32_t iterations; /* L-BFGS iterations K                                  */
    int32_t f_calls;    /* objective/gradient evaluations E                     */
    int32_t status;     /* MO_STATUS_*                                          */
    int32_t hist_words; /* sum_k min(k-1,m) over iterations (two-loop pairs)    */
    double f_min;       /* minimum of -logLike                                  */
    double gnorm;       /* ||grad||_inf at the returned point                   */
} mo_info;

/* ---------------------------------------------------------------- Philox4x32-10 */
/* Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11. */
void mo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* log(x) for finite x > 0: the classic fdlibm/musl argument reduction + degree-14 series. */
static double mo_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    uint32_t hx = (uint32_t)(bits >> 32);
    int k = 0;
    if (hx < 0x00100000u) { /* subnormal: scale up by 2^54 */
        k -= 54;
        x *= 18014398509481984.0;
        memcpy(&bits, &x, 8);
        hx = (uint32_t)(bits >> 32);
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    bits = ((uint64_t)hx << 32) | (bits & 0xffffffffu);
    memcpy(&x, &bits, 8);
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double dk = (double)k;
    return fma(dk, ln2_hi, (fma(s, hfsq + R, dk * ln2_lo) - hfsq) + f);
}

/* sin(pi t), cos(pi t) for t in [0,2): exact octant reduction, fdlibm kernels on |y|<=pi/4. */
static void mo_sincospi(double t, double* sn, double* cs) {
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
                        C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11,
                        PI = 3.14159265358979311600e+00;
    int n = (int)(2.0 * t + 0.5); /* nearest half-integer index, 0..4 */
    double r = t - 0.5 * (double)n; /* exact, |r| <= 1/4 */
    double y = r * PI;
    double z = y * y;
    double w = z * z;
    double rs = fma(z * w, fma(z, S6, S5), fma(z, fma(z, S4, S3), S2));
    double v = z * y;
    double ks = fma(v, fma(z, rs, S1), y);
    double rc = fma(w * w, fma(z, fma(z, C6, C5), C4), z * fma(z, fma(z, C3, C2), C1));
    double hz = 0.5 * z;
    double ww = 1.0 - hz;
    double kc = ww + fma(z, rc, (1.0 - ww) - hz);
    switch (n & 3) {
        case 0: *sn = ks; *cs = kc; break;
        case 1: *sn = kc; *cs = -ks; break;
        case 2: *sn = -ks; *cs = -kc; break;
        default: *sn = -kc; *cs = ks; break;
    }
}

/* Two independent standard normals for (seed, sim, element i). */
void mo_normal_pair(uint64_t seed, uint64_t sim, uint64_t i, double* n1, double* n2) {
    uint32_t ctr[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)sim, (uint32_t)(sim >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    mo_philox4x32_10(ctr, key, w);
    uint64_t k1 = ((uint64_t)w[0] << 20) | (w[1] >> 12);
    uint64_t k2 = ((uint64_t)w[2] << 20) | (w[3] >> 12);
    double u1 = ((double)k1 + 0.5) * 2.220446049250313080847e-16; /* 2^-52, in (0,1) */
    double u2 = ((double)k2 + 0.5) * 2.220446049250313080847e-16;
    double r = sqrt(-2.0 * mo_log(u1));
    double sn, cs;
    mo_sincospi(2.0 * u2, &sn, &cs);
    *n1 = r * cs;
    *n2 = r * sn;
}

void mo_normals(uint64_t seed, uint64_t sim, int64_t N, double* n1, double* n2) {
    for (int64_t i = 0; i < N; ++i) mo_normal_pair(seed, sim, (uint64_t)i, &n1[i], &n2[i]);
}

#ifdef MO_FAST
/* bench.py's second CPU figure (Makefile target `fast`; never the checker): the same generator written so that the
 * compiler can run it in SIMD lanes -- one loop over the elements, no call, no branch (the uniforms are >= 2^-53, so log's
 * subnormal path is never taken; the quadrant of sincospi by selects).  The same operations per element as mo_normal_pair,
 * free to be contracted and re-associated by this build's flags. */
static void mo_normals_simd(uint64_t seed, uint64_t sim, int64_t N, double* restrict n1o, double* restrict n2o) {
    const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32), s0 = (uint32_t)sim, s1 = (uint32_t)(sim >> 32);
#pragma omp simd
    for (int64_t i = 0; i < N; ++i) {
        uint32_t c0 = (uint32_t)i, c1 = (uint32_t)((uint64_t)i >> 32), c2 = s0, c3 = s1, k0 = key0, k1 = key1;
#pragma GCC unroll 10
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t m0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, m1 = (uint32_t)p1, m2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, m3 = (uint32_t)p0;
            c0 = m0; c1 = m1; c2 = m2; c3 = m3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const uint64_t q1 = ((uint64_t)c0 << 20) | (c1 >> 12), q2 = ((uint64_t)c2 << 20) | (c3 >> 12);
        const double u1 = ((double)(int64_t)q1 + 0.5) * 2.220446049250313080847e-16;
        const double u2 = ((double)(int64_t)q2 + 0.5) * 2.220446049250313080847e-16;
        /* log(u1), u1 in [2^-53, 1) */
        union { double d; uint64_t u; } b;
        b.d = u1;
        uint32_t hx = (uint32_t)(b.u >> 32) + (0x3ff00000u - 0x3fe6a09eu);
        const double dk = (double)((int)(hx >> 20) - 0x3ff);
        hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
        b.u = ((uint64_t)hx << 32) | (b.u & 0xffffffffu);
        const double f = b.d - 1.0, hfsq = 0.5 * f * f, sl = f / (2.0 + f), zl = sl * sl, wl = zl * zl;
        const double t1 = wl * (wl * (wl * 1.531383769920937332e-01 + 2.222219843214978396e-01) + 3.999999999940941908e-01);
        const double t2 = zl * (wl * (wl * (wl * 1.479819860511658591e-01 + 1.818357216161805012e-01) + 2.857142874366239149e-01) + 6.666666666666735130e-01);
        const double lg = dk * 6.93147180369123816490e-01 + ((sl * (hfsq + (t2 + t1)) + dk * 1.90821492927058770002e-10) - hfsq + f);
        const double rr = sqrt(-2.0 * lg);
        /* sincospi(2 u2) */
        const double t = 2.0 * u2;
        const int n = (int)(2.0 * t + 0.5);
        const double y = (t - 0.5 * (double)n) * 3.14159265358979311600e+00, z = y * y, w = z * z;
        const double rs = z * w * (z * 1.58969099521155010221e-10 + -2.50507602534068634195e-08) + (z * (z * 2.75573137070700676789e-06 + -1.98412698298579493134e-04) + 8.33333333332248946124e-03);
        const double ks = z * y * (z * rs + -1.66666666666666324348e-01) + y;
        const double rc = w * w * (z * (z * -1.13596475577881948265e-11 + 2.08757232129817482790e-09) + -2.75573143513906633035e-07)
                          + z * (z * (z * 2.48015872894767294178e-05 + -1.38888888888741095749e-03) + 4.16666666666666019037e-02);
        const double hz = 0.5 * z, ww = 1.0 - hz, kc = ww + (z * rc + ((1.0 - ww) - hz));
        const int qd = n & 3;
        const double sn = qd == 0 ? ks : (qd == 1 ? kc : (qd == 2 ? -ks : -kc));
        const double cs = qd == 0 ? kc : (qd == 1 ? -ks : (qd == 2 ? -kc : ks));
        n1o[i] = rr * cs;
        n2o[i] = rr * sn;
    }
}
#endif

/* ---------------------------------------------------------------- models */
static inline int mo_block(int64_t i, int64_t N, int B) { return (int)((i * (int64_t)B) / N); }

static inline double mo_Az(const double* z, int64_t i, int64_t N) { /* periodic (1/4,1/2,1/4) */
    int64_t im = (i == 0) ? N - 1 : i - 1, ip = (i == N - 1) ? 0 : i + 1;
    return fma(0.25, z[im] + z[ip], 0.5 * z[i]);
}


/* sum_i theta_k(i) = sum_k N_k theta_k, N_k = size of block k */
/* exp(theta/2) and exp(-theta) of the models.  The engine does not call a libm here: its device-resident muse! loop forms
 * the next theta's exponentials on the GPU, and host and device have to agree to the bit, so the engine defines exp as
 * one fixed sequence of IEEE operations -- fdlibm's e_exp.c: x = k ln2 + r with |r| <= ln2/2 (k = trunc(x / ln2 +- 1/2),
 * r = (x - k ln2_hi) - k ln2_lo), exp(r) = 1 - ((lo - r c / (2 - c)) - hi) with c = r - r^2 P(r^2) (Remez, degree 5 in
 * r^2), result scaled by 2^k.  Restated here from that published algorithm (not from the engine's source); the sampler
 * parity tests are bit-exact, so the two sides must evaluate the same sequence. */
static double mo_exp(double x) {
    static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10,
                        INV_LN2 = 1.44269504088896338700e+00;
    static const double P[5] = {1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05,
                                -1.65339022054652515390e-06, 4.13813679705723846039e-08};
    if (isnan(x)) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    int k = (int)(INV_LN2 * x + (x < 0.0 ? -0.5 : 0.5)); /* conversion truncates toward zero */
    double hi = x - (double)k * LN2_HI;
    double lo = (double)k * LN2_LO;
    double r = hi - lo;
    double r2 = r * r;
    double poly = P[4];
    for (int j = 3; j >= 0; --j) poly = P[j] + r2 * poly;
    double c = r - r2 * poly;
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021 && k <= 1023) return y * ldexp(1.0, k);      /* 2^k is a normal number: one exact scaling */
    if (k > 1023) return (y * 2.0) * ldexp(1.0, k - 1);
    return (y * ldexp(1.0, k + 1000)) * ldexp(1.0, -1000);       /* towards the subnormal range in two steps */
}

double mo_expd(double x) { return mo_exp(x); } /* for tests/test_oracle.py */

static double mo_theta_const(int64_t N, int ntheta, const double* theta) {
    double cst = 0.0;
    for (int k = 0; k < ntheta; ++k) {
        int64_t lo = ((int64_t)k * N + ntheta - 1) / ntheta, hi = ((int64_t)(k + 1) * N + ntheta - 1) / ntheta;
        cst += (double)(hi - lo) * theta[k];
    }
    return cst;
}

#if defined(MO_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
/* A header of the two-parameter family (include/muse_model.h, MUSE_MODEL_PAIR): K = ntheta / 2 blocks, block k's parameters
 * theta[k] and theta[K + k]; its four coefficients c[k][0..3] and the constant sum_k n_k C(a_k, b_k) come from the header. */
static double mo_pair_coefs(int64_t N, int ntheta, const double* theta, double c[][4]) {
    const int K = ntheta / 2;
    double cst = 0.0;
    for (int k = 0; k < K; ++k) {
        const int64_t lo = ((int64_t)k * N + K - 1) / K, hi = ((int64_t)(k + 1) * N + K - 1) / K;
        c[k][0] = c[k][1] = c[k][2] = c[k][3] = 0.0;
        cst += (double)(hi - lo) * muse_model_coefs(theta[k], theta[K + k], c[k]);
    }
    return cst;
}
#endif

/* sample_x_z(prob, rng, theta) -> (x, z)   [src/interface.jl:92-99, src/simple.jl:61-65] */
void mo_sample_x_z(int model, int64_t N, int ntheta, uint64_t seed, uint64_t sim, const double* theta,
                   double* x, double* z) {
    double sd[64];
    for (int k = 0; k < ntheta; ++k) sd[k] = mo_exp(0.5 * theta[k]);
#ifdef MO_USER_MODEL_HEADER
    if (model == MO_MODEL_USER) {
        MO_USER_CHECK_N(N);
#ifdef MUSE_MODEL_PAIR
        double c[32][4];
        mo_pair_coefs(N, ntheta, theta, c);
        for (int64_t i = 0; i < N; ++i) {
            double n1, n2;
            mo_normal_pair(seed, sim, (uint64_t)i, &n1, &n2);
            muse_model_sample(c[mo_block(i, N, ntheta / 2)], n1, n2, &z[i], &x[i], (long)i);
        }
#else
        for (int64_t i = 0; i < N; ++i) {
            double n1, n2;
            mo_normal_pair(seed, sim, (uint64_t)i, &n1, &n2);
            muse_model_sample(sd[mo_block(i, N, ntheta)], n1, n2, &z[i], &x[i], (long)i);
        }
#endif
        return;
    }
#endif
    if (model == MO_MODEL_NOISE) {
        for (int64_t i = 0; i < N; ++i) {
            double n1, n2;
            mo_normal_pair(seed, sim, (uint64_t)i, &n1, &n2);
            z[i] = n1;
            x[i] = n1 + sd[0] * n2;
        }
    } else if (model == MO_MODEL_FUNNEL) {
#ifdef MO_FAST
        mo_normals_simd(seed, sim, N, z, x);   /* (z <- n1, x <- n2) */
        for (int64_t i = 0; i < N; ++i) {
            z[i] = sd[mo_block(i, N, ntheta)] * z[i];
            x[i] = z[i] + x[i];
        }
#else
        for (int64_t i = 0; i < N; ++i) {
            double n1, n2;
            mo_normal_pair(seed, sim, (uint64_t)i, &n1, &n2);
            z[i] = sd[mo_block(i, N, ntheta)] * n1;
            x[i] = z[i] + n2;
        }
#endif
    } else { /* SMOOTH: x = A z + n2 */
        for (int64_t i = 0; i < N; ++i) {
            double n1, n2;
            mo_normal_pair(seed, sim, (uint64_t)i, &n1, &n2);
            z[i] = sd[mo_block(i, N, ntheta)] * n1;
            x[i] = n2;
        }
        for (int64_t i = 0; i < N; ++i) x[i] = mo_Az(z, i, N) + x[i];
    }
}

/* F(z) = -logLike(x,z,theta) and G = dF/dz.  This is what Optim minimises
 * (src/interface.jl:163: only_fg(z -> .-logLike_and_grad_z)). */
/* TEST-ONLY objective (model id 100): Rosenbrock's function, the example of the Optim.jl documentation
 * ("Minimizing a multivariate function"): f = (1 - x1)^2 + 100 (x2 - x1^2)^2 from x0 = (0, 0).  It lets the restated
 * LBFGS()/HagerZhang() be run on the one problem for which the package's own documentation prints the work counters. */
#define MO_MODEL_ROSENBROCK 100

double mo_negloglike_grad(int model, int64_t N, int ntheta, const double* x, const double* z,
                          const double* theta, double* G) {
    if (model == MO_MODEL_ROSENBROCK) {
        const double a = 1.0 - z[0], b = z[1] - z[0] * z[0];
        if (G) {
            G[0] = -2.0 * a - 400.0 * b * z[0];
            G[1] = 200.0 * b;
        }
        return a * a + 100.0 * b * b;
    }
    double iv[64];
    for (int k = 0; k < ntheta; ++k) iv[k] = mo_exp(-theta[k]);
    double acc = 0.0, cst = 0.0;
#ifdef MO_USER_MODEL_HEADER
    if (model == MO_MODEL_USER) {
        MO_USER_CHECK_N(N);
#ifdef MUSE_MODEL_PAIR
        double c[32][4];
        const double cst2 = mo_pair_coefs(N, ntheta, theta, c);
        for (int64_t i = 0; i < N; ++i) {
            const double gi = muse_model_grad(c[mo_block(i, N, ntheta / 2)], x[i], z[i], &acc, (long)i);
            if (G) G[i] = gi;
        }
        return 0.5 * (acc + cst2);
#else
        for (int64_t i = 0; i < N; ++i) {
            const double gi = muse_model_grad(iv[mo_block(i, N, ntheta)], x[i], z[i], &acc, (long)i);
            if (G) G[i] = gi;
        }
        return 0.5 * (acc + mo_theta_const(N, ntheta, theta));
#endif
    }
#endif
    if (model == MO_MODEL_NOISE) {
        for (int64_t i = 0; i < N; ++i) {
            double r = x[i] - z[i], t = iv[0] * r;
            acc = fma(z[i], z[i], fma(t, r, acc));
            if (G) G[i] = z[i] - t;
        }
        cst = (double)N * theta[0];
    } else if (model == MO_MODEL_FUNNEL) {
        for (int64_t i = 0; i < N; ++i) {
            int k = mo_block(i, N, ntheta);
            double r = x[i] - z[i], t = iv[k] * z[i];
            acc = fma(t, z[i], fma(r, r, acc));
            if (G) G[i] = t - r;
        }
        cst = mo_theta_const(N, ntheta, theta);
    } else {
        /* F = 1/2 |x - A z|^2 + 1/2 sum iv z^2 + const ; G = iv z - A^T (x - A z), A symmetric */
        double* r = (double*)malloc((size_t)N * sizeof(double));
        for (int64_t i = 0; i < N; ++i) r[i] = x[i] - mo_Az(z, i, N);
        for (int64_t i = 0; i < N; ++i) {
            int k = mo_block(i, N, ntheta);
            double t = iv[k] * z[i];
            acc = fma(t, z[i], fma(r[i], r[i], acc));
            if (G) G[i] = t - mo_Az(r, i, N);
        }
        free(r);
        cst = mo_theta_const(N, ntheta, theta);
    }
    return 0.5 * (acc + cst);
}

/* logLike_and_grad_z(prob,x,z,theta) -> (logLike, grad_z logLike)  [src/interface.jl:68-83] */
double mo_logLike_and_grad_z(int model, int64_t N, int ntheta, const double* x, const double* z,
                             const double* theta, double* g) {
    double F = mo_negloglike_grad(model, N, ntheta, x, z, theta, g);
    if (g)
        for (int64_t i = 0; i < N; ++i) g[i] = -g[i];
    return -F;
}

/* grad_theta logLike(x,z,theta)  [src/interface.jl:41-58, src/simple.jl:84] */
void mo_grad_theta(int model, int64_t N, int ntheta, const double* x, const double* z, const double* theta,
                   double* out) {
    if (model == MO_MODEL_NOISE) {
        double acc = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            double r = x[i] - z[i];
            acc += r * r;
        }
        out[0] = 0.5 * (mo_exp(-theta[0]) * acc - (double)N);
        return;
    }
    double acc[64];
    int64_t cnt[64];
    for (int k = 0; k < ntheta; ++k) { acc[k] = 0.0; cnt[k] = 0; }
#ifdef MO_USER_MODEL_HEADER
    if (model == MO_MODEL_USER) {
        MO_USER_CHECK_N(N);
#ifdef MUSE_MODEL_PAIR
        {   /* two block sums per block, the score's two components of a block assembled by the header */
            const int K = ntheta / 2;
            double c[32][4], s0[32], s1[32];
            mo_pair_coefs(N, ntheta, theta, c);
            for (int k = 0; k < K; ++k) s0[k] = s1[k] = 0.0;
            for (int64_t i = 0; i < N; ++i) {
                int k = mo_block(i, N, K);
                double t0, t1;
                muse_model_score_terms(c[k], x[i], z[i], &t0, &t1, (long)i);
                s0[k] += t0;
                s1[k] += t1;
                cnt[k] += 1;
            }
            for (int k = 0; k < K; ++k) muse_model_score(c[k], s0[k], s1[k], (double)cnt[k], &out[k], &out[K + k]);
            return;
        }
#else
        for (int64_t i = 0; i < N; ++i) {
            int k = mo_block(i, N, ntheta);
            acc[k] += muse_model_score_term(x[i], z[i], (long)i);
            cnt[k] += 1;
        }
#endif
    } else
#endif
    for (int64_t i = 0; i < N; ++i) {
        int k = mo_block(i, N, ntheta);
        acc[k] += z[i] * z[i];
        cnt[k] += 1;
    }
    for (int k = 0; k < ntheta; ++k) out[k] = 0.5 * (mo_exp(-theta[k]) * acc[k] - (double)cnt[k]);
}

/* ---------------------------------------------------------------- objective wrapper with
 * the NLSolversBase value_gradient! cache (evaluation is skipped when the point is the one
 * last evaluated), which decides f_calls. */
typedef struct {
    int model, ntheta;
    int64_t N;
    const double *x, *theta;
    double* x_last; /* point of the last evaluation */
    double* g;      /* gradient(d): gradient at x_last */
    double f;       /* value(d) */
    int have;
    int f_calls;
} mo_obj;

static void mo_value_gradient(mo_obj* d, const double* xp) {
    if (d->have && memcmp(d->x_last, xp, (size_t)d->N * sizeof(double)) == 0) return;
    d->f = mo_negloglike_grad(d->model, d->N, d->ntheta, d->x, xp, d->theta, d->g);
    memcpy(d->x_last, xp, (size_t)d->N * sizeof(double));
    d->have = 1;
    d->f_calls += 1;
}

static double mo_dot(const double* a, const double* b, int64_t N) {
    double s = 0.0;
    for (int64_t i = 0; i < N; ++i) s = fma(a[i], b[i], s);
    return s;
}
static double mo_maxabs(const double* a, int64_t N) {
    double m = 0.0;
    for (int64_t i = 0; i < N; ++i) {
        double v = fabs(a[i]);
        if (!(v <= m)) m = v; /* propagates NaN like Julia's maximum(abs, .) */
    }
    return m;
}
static int mo_allfinite(const double* a, int64_t N) {
    for (int64_t i = 0; i < N; ++i)
        if (!isfinite(a[i])) return 0;
    return 1;
}

/* ---------------------------------------------------------------- HagerZhang line search */
typedef struct {
    double *alphas, *values, *slopes;
    int n, cap;
} hz_trace;
static void hz_push(hz_trace* t, double a, double v, double s) {
    if (t->n == t->cap) {
        t->cap *= 2;
        t->alphas = (double*)realloc(t->alphas, (size_t)t->cap * sizeof(double));
        t->values = (double*)realloc(t->values, (size_t)t->cap * sizeof(double));
        t->slopes = (double*)realloc(t->slopes, (size_t)t->cap * sizeof(double));
    }
    t->alphas[t->n] = a; t->values[t->n] = v; t->slopes[t->n] = s;
    t->n += 1;
}

typedef struct {
    mo_obj* d;
    const double *x, *s;
    double* x_new;
} hz_line;

/* phi(alpha), dphi(alpha): x_new = x + alpha s; value_gradient!(d, x_new). */
static void hz_phidphi(hz_line* L, double alpha, double* phi, double* dphi) {
    int64_t N = L->d->N;
    for (int64_t i = 0; i < N; ++i) L->x_new[i] = fma(alpha, L->s[i], L->x[i]);
    mo_value_gradient(L->d, L->x_new);
    *phi = L->d->f;
    *dphi = mo_dot(L->d->g, L->s, N);
}

#define HZ_DELTA 0.1
#define HZ_SIGMA 0.9
#define HZ_RHO 5.0
#define HZ_EPSILON 1e-6
#define HZ_GAMMA 0.66
#define HZ_LINESEARCHMAX 50
#define HZ_PSI3 0.1
#define MO_EPS 2.220446049250313e-16

static double hz_eps_of(double b) { return nextafter(fabs(b), INFINITY) - fabs(b); } /* eps(b) */

static int hz_satisfies_wolfe(double c, double phi_c, double dphi_c, double phi_0, double dphi_0,
                              double phi_lim) {
    int wolfe1 = (HZ_DELTA * dphi_0 >= (phi_c - phi_0) / c) && (dphi_c >= HZ_SIGMA * dphi_0);
    int wolfe2 = ((2.0 * HZ_DELTA - 1.0) * dphi_0 >= dphi_c) && (dphi_c >= HZ_SIGMA * dphi_0) &&
                 (phi_c <= phi_lim);
    return wolfe1 || wolfe2;
}

/* HZ stage U3 (theta = 1/2).  Returns 0 ok, -1 on a failed internal assertion. */
static int hz_bisect(hz_line* L, hz_trace* t, int ia, int ib, double phi_lim, int* oa, int* ob) {
    double a = t->alphas[ia], b = t->alphas[ib];
    if (!(t->slopes[ia] < 0.0 && t->values[ia] <= phi_lim && t->slopes[ib] < 0.0 &&
          t->values[ib] > phi_lim && b > a))
        return -1;
    while (b - a > hz_eps_of(b)) {
        double d = (a + b) / 2.0, phi_d, gphi;
        hz_phidphi(L, d, &phi_d, &gphi);
        if (!(isfinite(phi_d) && isfinite(gphi))) return -1;
        hz_push(t, d, phi_d, gphi);
        int id = t->n - 1;
        if (gphi >= 0.0) { *oa = ia; *ob = id; return 0; }
        if (phi_d <= phi_lim) { a = d; ia = id; }
        else { b = d; ib = id; }
    }
    *oa = ia; *ob = ib;
    return 0;
}

/* HZ stages U0-U3. */
static int hz_update(hz_line* L, hz_trace* t, int ia, int ib, int ic, double phi_lim, int* oa, int* ob) {
    double a = t->alphas[ia], b = t->alphas[ib];
    if (!(t->slopes[ia] < 0.0 && t->values[ia] <= phi_lim && t->slopes[ib] >= 0.0 && b > a)) return -1;
    double c = t->alphas[ic], phi_c = t->values[ic], dphi_c = t->slopes[ic];
    if (c < a || c > b) { *oa = ia; *ob = ib; return 0; }
    if (dphi_c >= 0.0) { *oa = ia; *ob = ic; return 0; }
    if (phi_c <= phi_lim) { *oa = ic; *ob = ib; return 0; }
    return hz_bisect(L, t, ia, ic, phi_lim, oa, ob);
}

static double hz_secant(double a, double b, double dphi_a, double dphi_b) {
    return (a * dphi_b - b * dphi_a) / (dphi_b - dphi_a);
}

/* HZ stages S1-S4.  Returns 1 wolfe, 0 not, -1 failure. */
static int hz_secant2(hz_line* L, hz_trace* t, int ia, int ib, double phi_lim, int* oA, int* oB) {
    double phi_0 = t->values[0], dphi_0 = t->slopes[0];
    double a = t->alphas[ia], b = t->alphas[ib];
    double dphi_a = t->slopes[ia], dphi_b = t->slopes[ib];
    if (!(dphi_a < 0.0 && dphi_b >= 0.0)) return -1;
    double c = hz_secant(a, b, dphi_a, dphi_b);
    if (!isfinite(c)) return -1;
    double phi_c, dphi_c;
    hz_phidphi(L, c, &phi_c, &dphi_c);
    if (!(isfinite(phi_c) && isfinite(dphi_c))) return -1;
    hz_push(t, c, phi_c, dphi_c);
    int ic = t->n - 1;
    if (hz_satisfies_wolfe(c, phi_c, dphi_c, phi_0, dphi_0, phi_lim)) { *oA = ic; *oB = ic; return 1; }
    int iA, iB;
    if (hz_update(L, t, ia, ib, ic, phi_lim, &iA, &iB) < 0) return -1;
    a = t->alphas[iA];
    b = t->alphas[iB];
    if (iB == ic) c = hz_secant(t->alphas[ib], t->alphas[iB], t->slopes[ib], t->slopes[iB]);
    else if (iA == ic) c = hz_secant(t->alphas[ia], t->alphas[iA], t->slopes[ia], t->slopes[iA]);
    if ((iA == ic || iB == ic) && a <= c && c <= b) {
        hz_phidphi(L, c, &phi_c, &dphi_c);
        if (!(isfinite(phi_c) && isfinite(dphi_c))) return -1;
        hz_push(t, c, phi_c, dphi_c);
        ic = t->n - 1;
        if (hz_satisfies_wolfe(c, phi_c, dphi_c, phi_0, dphi_0, phi_lim)) { *oA = ic; *oB = ic; return 1; }
        int jA, jB;
        if (hz_update(L, t, iA, iB, ic, phi_lim, &jA, &jB) < 0) return -1;
        iA = jA; iB = jB;
    }
    *oA = iA; *oB = iB;
    return 0;
}

/* The line search proper.  Returns 0 and *alpha on success; -1 (LineSearchException or a
 * failed assertion) with *alpha = the step Optim would fall back to. */
static int hz_linesearch(hz_line* L, double c, double phi_0, double dphi_0, double* alpha) {
    double alphamax = INFINITY;
    *alpha = 0.0;
    if (!(isfinite(phi_0) && isfinite(dphi_0))) return -1;
    if (dphi_0 >= MO_EPS * fabs(phi_0)) return -1; /* not a descent direction */
    if (dphi_0 >= 0.0) return 0;                    /* alpha = 0 */
    const int iterfinitemax = 52;                   /* ceil(-log2(eps)) */
    hz_trace t;
    t.cap = 16; t.n = 0;
    t.alphas = (double*)malloc(16 * sizeof(double));
    t.values = (double*)malloc(16 * sizeof(double));
    t.slopes = (double*)malloc(16 * sizeof(double));
    hz_push(&t, 0.0, phi_0, dphi_0);
    int rc = -1;
    double phi_lim = phi_0 + HZ_EPSILON * fabs(phi_0);
    if (c <= MO_EPS) { rc = 0; goto done; }
    double phi_c, dphi_c;
    hz_phidphi(L, c, &phi_c, &dphi_c);
    int iterfinite = 1;
    while (!(isfinite(phi_c) && isfinite(dphi_c)) && iterfinite < iterfinitemax) {
        iterfinite += 1;
        c *= HZ_PSI3;
        hz_phidphi(L, c, &phi_c, &dphi_c);
    }
    if (!(isfinite(phi_c) && isfinite(dphi_c))) { rc = 0; goto done; } /* alpha = 0 */
    hz_push(&t, c, phi_c, dphi_c);
    /* mayterminate is false under InitialStatic: no Wolfe test on the initial point. */
    int isbracketed = 0, ia = 0, ib = 1, iter = 1;
    double cold = -1.0;
    while (!isbracketed && iter < HZ_LINESEARCHMAX) {
        if (dphi_c >= 0.0) {
            ib = t.n - 1;
            for (int i = ib - 1; i >= 0; --i)
                if (t.values[i] <= phi_lim) { ia = i; break; }
            isbracketed = 1;
        } else if (t.values[t.n - 1] > phi_lim) {
            ib = t.n - 1;
            ia = 0;
            if (hz_bisect(L, &t, ia, ib, phi_lim, &ia, &ib) < 0) { *alpha = t.alphas[ia]; goto done; }
            isbracketed = 1;
        } else {
            cold = c;
            if (nextafter(cold, INFINITY) >= alphamax) { *alpha = cold; rc = 0; goto done; }
            c *= HZ_RHO;
            if (c > alphamax) c = alphamax;
            hz_phidphi(L, c, &phi_c, &dphi_c);
            iterfinite = 1;
            while (!(isfinite(phi_c) && isfinite(dphi_c)) && c > nextafter(cold, INFINITY) &&
                   iterfinite < iterfinitemax) {
                alphamax = c;
                iterfinite += 1;
                c = (cold + c) / 2.0;
                hz_phidphi(L, c, &phi_c, &dphi_c);
            }
            if (!(isfinite(phi_c) && isfinite(dphi_c))) { *alpha = cold; rc = 0; goto done; }
            hz_push(&t, c, phi_c, dphi_c);
        }
        iter += 1;
    }
    while (iter < HZ_LINESEARCHMAX) {
        double a = t.alphas[ia], b = t.alphas[ib];
        if (!(b > a)) { *alpha = a; goto done; }
        if (b - a <= hz_eps_of(b)) { *alpha = a; rc = 0; goto done; }
        int iA, iB;
        int w = hz_secant2(L, &t, ia, ib, phi_lim, &iA, &iB);
        if (w < 0) { *alpha = a; goto done; }
        if (w == 1) { *alpha = t.alphas[iA]; rc = 0; goto done; }
        double A = t.alphas[iA], B = t.alphas[iB];
        if (!(B > A)) { *alpha = A; goto done; }
        if (B - A < HZ_GAMMA * (b - a)) {
            if (nextafter(t.values[ia], INFINITY) >= t.values[ib] &&
                nextafter(t.values[iA], INFINITY) >= t.values[iB]) {
                *alpha = A; rc = 0; goto done; /* flat: secant did nothing useful */
            }
            ia = iA; ib = iB;
        } else {
            c = (A + B) / 2.0;
            hz_phidphi(L, c, &phi_c, &dphi_c);
            if (!(isfinite(phi_c) && isfinite(dphi_c))) { *alpha = A; goto done; }
            hz_push(&t, c, phi_c, dphi_c);
            if (hz_update(L, &t, iA, iB, t.n - 1, phi_lim, &ia, &ib) < 0) { *alpha = A; goto done; }
        }
        iter += 1;
    }
    *alpha = t.alphas[ia]; /* LineSearchException: reached linesearchmax */
done:
    free(t.alphas); free(t.values); free(t.slopes);
    return rc;
}

/* ---------------------------------------------------------------- Optim LBFGS */
#define LBFGS_M 10
#define LBFGS_MAXITER 1000
static inline int mod1(int i, int m) { int r = i % m; return r == 0 ? m : r; }

/* zhat_at_theta(prob, x, z0, theta; atol) -> (zhat, info)   [src/interface.jl:162-171] */
int mo_zhat_at_theta(int model, int64_t N, int ntheta, const double* x, const double* z0,
                     const double* theta, double atol, double* zout, mo_info* info) {
    size_t nb = (size_t)N * sizeof(double);
    const int outermost = (tl_off == 0);
    if (outermost) ws_reserve(N);
    const size_t mark = tl_off;
    mo_obj d;
    d.model = model; d.ntheta = ntheta; d.N = N; d.x = x; d.theta = theta;
    d.x_last = ws_alloc(N); d.g = ws_alloc(N);
    d.have = 0; d.f_calls = 0; d.f = 0.0;
    double* X = ws_alloc(N);
    double* Xprev = ws_alloc(N);
    double* gprev = ws_alloc(N);
    double* s = ws_alloc(N);
    double* q = ws_alloc(N);
    double* dx = ws_alloc(N);
    double* dg = ws_alloc(N);
    double* x_ls = ws_alloc(N);
    double* dxh = ws_alloc((size_t)N * LBFGS_M);
    double* dgh = ws_alloc((size_t)N * LBFGS_M);
    double rho[LBFGS_M + 1], tl_alpha[LBFGS_M + 1];
    memcpy(X, z0, nb);

    /* initial_state: value_gradient!!(d, x0); initial_convergence */
    mo_value_gradient(&d, X);
    int iterations = 0, pseudo = 0, status = MO_STATUS_MAXITER, hist_words = 0;
    int counter_f_tol = 0;
    double f_prev = NAN;
    int converged = 0;
    if (!isfinite(d.f) || !mo_allfinite(d.g, N)) {
        status = MO_STATUS_NONFINITE;
        converged = 1; /* "stopped": loop not entered */
    } else if (mo_maxabs(d.g, N) <= atol) {
        status = MO_STATUS_G_CONVERGED;
        converged = 1;
    }
    while (!converged && iterations < LBFGS_MAXITER) {
        iterations += 1;
        /* ---- update_state! ---- */
        pseudo += 1;
        {   /* twoloop! */
            int lower = pseudo - LBFGS_M, upper = pseudo - 1;
            memcpy(q, d.g, nb);
            for (int index = upper; index >= lower; --index) {
                if (index < 1) continue;
                int i = mod1(index, LBFGS_M);
                const double *dgi = dgh + (size_t)(i - 1) * N, *dxi = dxh + (size_t)(i - 1) * N;
                tl_alpha[i] = rho[i] * mo_dot(dxi, q, N);
                for (int64_t e = 0; e < N; ++e) q[e] = fma(-tl_alpha[i], dgi[e], q[e]);
                hist_words += 1;
            }
            if (pseudo > 1) { /* scaleinvH0: Nocedal & Wright eq. (7.20) */
                int i = mod1(upper, LBFGS_M);
                const double *dgi = dgh + (size_t)(i - 1) * N, *dxi = dxh + (size_t)(i - 1) * N;
                double scaling = mo_dot(dxi, dgi, N) / mo_dot(dgi, dgi, N);
                for (int64_t e = 0; e < N; ++e) s[e] = scaling * q[e];
            } else {
                memcpy(s, q, nb);
            }
            for (int index = lower; index <= upper; ++index) {
                if (index < 1) continue;
                int i = mod1(index, LBFGS_M);
                const double *dgi = dgh + (size_t)(i - 1) * N, *dxi = dxh + (size_t)(i - 1) * N;
                double beta = rho[i] * mo_dot(dgi, s, N);
                { const double coef = tl_alpha[i] - beta; for (int64_t e = 0; e < N; ++e) s[e] = fma(dxi[e], coef, s[e]); }
            }
            for (int64_t e = 0; e < N; ++e) s[e] *= -1.0;
        }
        memcpy(gprev, d.g, nb);
        /* perform_linesearch! */
        double dphi_0 = mo_dot(d.g, s, N);
        if (dphi_0 >= 0.0) { /* reset_search_direction! */
            pseudo = 1;
            for (int64_t e = 0; e < N; ++e) s[e] = -d.g[e];
            dphi_0 = mo_dot(d.g, s, N);
        }
        double phi_0 = d.f;
        f_prev = phi_0;
        memcpy(Xprev, X, nb);
        hz_line L;
        L.d = &d; L.x = X; L.s = s; L.x_new = x_ls;
        double alpha;
        int lsrc = hz_linesearch(&L, 1.0 /* InitialStatic */, phi_0, dphi_0, &alpha);
        for (int64_t e = 0; e < N; ++e) dx[e] = alpha * s[e];
        for (int64_t e = 0; e < N; ++e) X[e] = fma(alpha, s[e], X[e]); /* = the point the line search evaluated */
        if (lsrc < 0) { status = MO_STATUS_LINESEARCH_FAILED; break; }
        /* ---- update_g! ---- */
        mo_value_gradient(&d, X);
        /* ---- assess_convergence ---- */
        double xchange = 0.0;
        for (int64_t e = 0; e < N; ++e) {
            double v = fabs(X[e] - Xprev[e]);
            if (!(v <= xchange)) xchange = v;
        }
        int x_conv = (xchange <= 0.0);
        int f_conv = (fabs(d.f - f_prev) <= 0.0);
        int g_conv = (mo_maxabs(d.g, N) <= atol);
        counter_f_tol = f_conv ? counter_f_tol + 1 : 0;
        converged = x_conv || g_conv || (counter_f_tol > 1);
        if (g_conv) status = MO_STATUS_G_CONVERGED;
        else if (x_conv) status = MO_STATUS_X_CONVERGED;
        else if (counter_f_tol > 1) status = MO_STATUS_F_CONVERGED;
        /* ---- update_h! ---- */
        for (int64_t e = 0; e < N; ++e) dg[e] = d.g[e] - gprev[e];
        double rho_it = 1.0 / mo_dot(dx, dg, N);
        if (isinf(rho_it)) {
            pseudo = 0;
        } else {
            int idx = mod1(pseudo, LBFGS_M);
            memcpy(dxh + (size_t)(idx - 1) * N, dx, nb);
            memcpy(dgh + (size_t)(idx - 1) * N, dg, nb);
            rho[idx] = rho_it;
        }
        if (!mo_allfinite(d.g, N)) { status = MO_STATUS_NONFINITE; break; }
    }
    memcpy(zout, X, nb);
    if (info) {
        info->iterations = iterations;
        info->f_calls = d.f_calls;
        info->status = status;
        info->hist_words = hist_words;
        /* Optim reports value(d) at the last evaluated point; after update_g! that is X. */
        info->f_min = d.f;
        info->gnorm = mo_maxabs(d.g, N);
        if (status <= MO_STATUS_F_CONVERGED && !isfinite(d.f)) info->status = MO_STATUS_NONFINITE;
    }
    tl_off = mark;
    return 0;
}

/* ---------------------------------------------------------------- batched map bodies */
/* One element of the muse!/get_J! map (src/muse.jl:169-176, :508-525):
 *   x = data | sample_x_z(rng_sim, theta_sample).x ; zhat = zhat_at_theta(x, z0, theta) ;
 *   g = grad_theta(x, zhat, theta).
 * z0_mode: 0 zeros, 1 the simulation's true z (get_J!, src/muse.jl:511), 2 z_inout as given. */
int mo_map_and_score(int model, int64_t N, int ntheta, uint64_t seed, int64_t sim, const double* x_data,
                     const double* theta_sample, const double* theta, double atol, int z0_mode,
                     double* z_inout, double* g_out, mo_info* info) {
    size_t nb = (size_t)N * sizeof(double);
    const int outermost = (tl_off == 0);
    if (outermost) ws_reserve(N);
    const size_t mark = tl_off;
    double* x = ws_alloc(N);
    double* z0 = ws_alloc(N);
    if (sim < 0) {
        memcpy(x, x_data, nb);
        if (z0_mode == 2) memcpy(z0, z_inout, nb);
        else memset(z0, 0, nb);
    } else {
        mo_sample_x_z(model, N, ntheta, seed, (uint64_t)sim, theta_sample, x, z0);
        if (z0_mode == 0) memset(z0, 0, nb);
        else if (z0_mode == 2) memcpy(z0, z_inout, nb);
    }
    mo_zhat_at_theta(model, N, ntheta, x, z0, theta, atol, z_inout, info);
    mo_grad_theta(model, N, ntheta, x, z_inout, theta, g_out);
    tl_off = mark;
    return 0;
}

/* Batch over sims [sim_begin, sim_end) (+ the data element first when include_data).
 * zhat: [(n)][N] in/out (used as start when z0_mode==2), g_out: [(n)][ntheta]. */
int mo_map_and_score_batch(int model, int64_t N, int ntheta, uint64_t seed, int64_t sim_begin,
                           int64_t sim_end, int include_data, const double* x_data, const double* theta,
                           double atol, int z0_mode, double* zhat, double* g_out, mo_info* info,
                           int nthreads) {
    int64_t n = (sim_end - sim_begin) + (include_data ? 1 : 0);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t p = 0; p < n; ++p) {
        int64_t sim = include_data ? (p == 0 ? -1 : sim_begin + p - 1) : sim_begin + p;
        mo_map_and_score(model, N, ntheta, seed, sim, x_data, theta, theta, atol, z0_mode,
                         zhat + (size_t)p * N, g_out + (size_t)p * ntheta, info ? &info[p] : NULL);
    }
    (void)nthreads;
    return 0;
}

/* get_H! finite-difference branch for one sim (src/muse.jl:426-433, src/util.jl:9-27):
 * column j = (-1/2 f(theta0 - h_j e_j) + 0 f(theta0) + 1/2 f(theta0 + h_j e_j)) / h_j with
 * f(theta) = grad_theta(x(theta; same randoms), zhat(x; theta0, start zfid), theta0).
 * The centre evaluation has coefficient 0 and is skipped.  H_out is [ntheta][ntheta]
 * row-major with H_out[i][j] = d score_i / d theta_j (hcat of columns, src/util.jl:25). */
int mo_fd_jacobian(int model, int64_t N, int ntheta, uint64_t seed, int64_t sim, const double* theta0,
                   const double* step, double atol, const double* zfid, double* H_out) {
    size_t nb = (size_t)N * sizeof(double);
    double* z = (double*)malloc(nb);
    double th[64], gp[64], gm[64];
    for (int j = 0; j < ntheta; ++j) {
        for (int k = 0; k < ntheta; ++k) th[k] = theta0[k];
        th[j] = theta0[j] + step[j];
        memcpy(z, zfid, nb);
        mo_map_and_score(model, N, ntheta, seed, sim, NULL, th, theta0, atol, 2, z, gp, NULL);
        th[j] = theta0[j] + (-step[j]);
        memcpy(z, zfid, nb);
        mo_map_and_score(model, N, ntheta, seed, sim, NULL, th, theta0, atol, 2, z, gm, NULL);
        for (int i = 0; i < ntheta; ++i) H_out[i * ntheta + j] = (-0.5 * gm[i] + 0.5 * gp[i]) / step[j];
    }
    free(z);
    return 0;
}

/* ---------------------------------------------------------------- get_H! implicit-differentiation branch
 * (src/muse.jl:335-405), per sim:
 *   (x, z) = sample_x_z(rng, theta0); zhat = zhat_at_theta(x, zero(z), theta0; atol)   [the reference hard-codes
 *   atol = 1e-1 here, src/muse.jl:344; the caller passes it]
 *   H1     = d/dtheta [ grad_theta' logLike(x(theta), zhat, theta') ]_{theta'=theta0}          (:353-358)
 *   dFdth  = d/dtheta [ grad_z logLike(x, zhat, theta) ]            (N x ntheta)                 (:361-365)
 *   dFdth1 = d/dtheta [ grad_z logLike(x(theta), zhat, theta0) ]    (N x ntheta)                 (:366-371)
 *   A w    = Hessian_z logLike(x, zhat, theta0) w                                                (:373-379)
 *   H      = H1 - dFdth^T A^{-1} dFdth1, A^{-1} by conjugate gradients (IterativeSolvers.cg,
 *            maxiter = 100, reltol = sqrt(eps), abstol = 0, x0 = 0)                              (:380-389)
 * The reference obtains the derivatives by nested AD; for the compiled-in models they are closed forms:
 *   funnel/smooth: grad_z = A^T(x - A z) - iv_k z;  score_k = 1/2 (iv_k sum_k z^2 - N_k)  (no x dependence => H1 = 0)
 *                  dFdth[:,k] = iv_k zhat 1_k;  dx/dtheta_k = A (1/2 z_true 1_k);  dFdth1[:,k] = A^T A (1/2 z_true 1_k)
 *                  A_hess w = -(A^T A w + iv w)
 *   noise:         grad_z = iv (x - z) - z;  score = 1/2 (iv sum (x-z)^2 - N);  dx/dtheta = 1/2 (x - z_true)
 *                  H1 = iv sum (x - zhat) 1/2 (x - z_true);  dFdth = -iv (x - zhat);  dFdth1 = iv 1/2 (x - z_true)
 *                  A_hess w = -(iv + 1) w
 * H_out[i][j] row-major; cg_iters_out[ntheta] (may be NULL). */
static void mo_hess_apply(int model, int64_t N, int ntheta, const double* iv, const double* w, double* out, double* tmp) {
    if (model == MO_MODEL_USER) { /* tmp holds the elements' d2 o / dz2 (mo_implicit_H) */
        for (int64_t i = 0; i < N; ++i) out[i] = -(tmp[i] * w[i]);
    } else if (model == MO_MODEL_NOISE) {
        for (int64_t i = 0; i < N; ++i) out[i] = -((iv[0] + 1.0) * w[i]);
    } else if (model == MO_MODEL_FUNNEL) {
        for (int64_t i = 0; i < N; ++i) out[i] = -(w[i] + iv[mo_block(i, N, ntheta)] * w[i]);
    } else {
        for (int64_t i = 0; i < N; ++i) tmp[i] = mo_Az(w, i, N);
        for (int64_t i = 0; i < N; ++i) out[i] = -(mo_Az(tmp, i, N) + iv[mo_block(i, N, ntheta)] * w[i]);
    }
}

int mo_implicit_H(int model, int64_t N, int ntheta, uint64_t seed, int64_t sim, const double* theta0, double atol,
                  int cg_maxiter, double* H_out, int32_t* cg_iters_out) {
    size_t nb = (size_t)N * sizeof(double);
    double *x = (double*)malloc(nb), *zt = (double*)malloc(nb), *zh = (double*)malloc(nb), *z0 = (double*)calloc((size_t)N, sizeof(double));
    double *b = (double*)malloc(nb), *v = (double*)malloc(nb), *r = (double*)malloc(nb), *pp = (double*)malloc(nb);
    double *Ap = (double*)malloc(nb), *tmp = (double*)malloc(nb);
    double iv[64];
    for (int k = 0; k < ntheta; ++k) iv[k] = mo_exp(-theta0[k]);
    mo_sample_x_z(model, N, ntheta, seed, (uint64_t)sim, theta0, x, zt);
    mo_zhat_at_theta(model, N, ntheta, x, z0, theta0, atol, zh, NULL);
    /* a user-supplied model (include/muse_model.h, MUSE_MODEL_SECOND): the operands from the header's second derivatives --
     *   tx_i = dx_i/dtheta_k = 1/2 sd_k dx/dsd (kept in zt's place), A_hess w = -ozz w, dFdth1[:,k] = -ozx tx 1_k,
     *   dFdth[:,k] = 1/2 iv_k bz 1_k, H1[k][k] = 1/2 iv_k sum_k bx tx */
    double *uz = NULL, *ux = NULL, *ubz = NULL, *ubx = NULL;
    if (model == MO_MODEL_USER) {
#if defined(MO_USER_MODEL_HEADER) && defined(MUSE_MODEL_SECOND)
        uz = (double*)malloc(nb); ux = (double*)malloc(nb); ubz = (double*)malloc(nb); ubx = (double*)malloc(nb);
        for (int64_t i = 0; i < N; ++i) {
            const int k = mo_block(i, N, ntheta);
            const double sdk = mo_exp(0.5 * theta0[k]);
            double n1, n2;
            mo_normal_pair(seed, (uint64_t)sim, (uint64_t)i, &n1, &n2);
            zt[i] = 0.5 * (sdk * muse_model_dx_dsd(sdk, n1, n2, (long)i));
            muse_model_second(iv[k], x[i], zh[i], &uz[i], &ux[i], &ubz[i], &ubx[i], (long)i);
        }
#else
        free(x); free(zt); free(zh); free(z0); free(b); free(v); free(r); free(pp); free(Ap); free(tmp);
        return -1; /* the header has no second derivatives */
#endif
    }
    for (int j = 0; j < ntheta; ++j) {
        /* right-hand side dFdth1[:, j] */
        if (model == MO_MODEL_USER) {
            for (int64_t i = 0; i < N; ++i) b[i] = mo_block(i, N, ntheta) == j ? -(ux[i] * zt[i]) : 0.0;
            memcpy(tmp, uz, nb);
        } else if (model == MO_MODEL_NOISE) {
            for (int64_t i = 0; i < N; ++i) b[i] = iv[0] * (0.5 * (x[i] - zt[i]));
        } else {
            for (int64_t i = 0; i < N; ++i) tmp[i] = mo_block(i, N, ntheta) == j ? 0.5 * zt[i] : 0.0;
            if (model == MO_MODEL_FUNNEL) memcpy(b, tmp, nb);
            else {
                for (int64_t i = 0; i < N; ++i) r[i] = mo_Az(tmp, i, N);
                for (int64_t i = 0; i < N; ++i) b[i] = mo_Az(r, i, N);
            }
        }
        /* conjugate gradients on A v = b, x0 = 0 */
        memset(v, 0, nb);
        memcpy(r, b, nb);
        memcpy(pp, b, nb);
        double rr = mo_dot(r, r, N);
        const double tol = sqrt(MO_EPS) * sqrt(rr);
        int it = 0;
        while (it < cg_maxiter && !(sqrt(rr) <= tol)) {
            mo_hess_apply(model, N, ntheta, iv, pp, Ap, tmp);
            const double alpha = rr / mo_dot(pp, Ap, N);
            for (int64_t i = 0; i < N; ++i) v[i] = fma(alpha, pp[i], v[i]);
            for (int64_t i = 0; i < N; ++i) r[i] = fma(-alpha, Ap[i], r[i]);
            const double rr_new = mo_dot(r, r, N);
            const double beta = rr_new / rr;
            for (int64_t i = 0; i < N; ++i) pp[i] = fma(beta, pp[i], r[i]);
            rr = rr_new;
            it += 1;
        }
        if (cg_iters_out) cg_iters_out[j] = it;
        /* H[:, j] = H1[:, j] - dFdth^T v */
        for (int i2 = 0; i2 < ntheta; ++i2) {
            double acc = 0.0, h1 = 0.0;
            if (model == MO_MODEL_USER) {
                for (int64_t i = 0; i < N; ++i) {
                    const int k = mo_block(i, N, ntheta);
                    if (k == i2) acc = fma(0.5 * (iv[i2] * ubz[i]), v[i], acc);
                    if (k == j && i2 == j) h1 = fma(ubx[i], zt[i], h1);
                }
                h1 = 0.5 * (iv[i2] * h1);
            } else if (model == MO_MODEL_NOISE) {
                for (int64_t i = 0; i < N; ++i) {
                    const double d = x[i] - zh[i];
                    acc = fma(-iv[0] * d, v[i], acc);
                    h1 = fma(d, 0.5 * (x[i] - zt[i]), h1);
                }
                h1 *= iv[0];
            } else {
                for (int64_t i = 0; i < N; ++i)
                    if (mo_block(i, N, ntheta) == i2) acc = fma(iv[i2] * zh[i], v[i], acc);
            }
            H_out[i2 * ntheta + j] = h1 - acc;
        }
    }
    free(x); free(zt); free(zh); free(z0); free(b); free(v); free(r); free(pp); free(Ap); free(tmp);
    free(uz); free(ux); free(ubz); free(ubx);
    return 0;
}

int mo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
