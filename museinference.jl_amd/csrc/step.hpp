// step.hpp -- the scalar algebra between two maps of the muse! outer loop (src/muse.jl:163-166, 177-232), written ONCE for
// the host (muse_run's loop in muse_engine.cpp) and for the device (the step between two iterations of the device-resident
// loop, muse_loop_kernel in muse_kernels.hip): the same statements in the same order, compiled without floating-point contraction on both sides,
// every operation an IEEE +, -, *, / or sqrt -- so the two loops produce the same bits.  Also muse_exp, the exponential
// behind ThetaSet::sd / ::iv: a fixed fdlibm-style sequence instead of the host's libm, for the same reason (the device
// loop forms the next theta's exp(theta/2), exp(-theta) itself).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "args.hpp"

#if defined(__HIPCC__)
#include "reduce.hpp"
#define MUSE_HD __host__ __device__ inline
#else
#define MUSE_HD inline
#endif

namespace muse {

// exp(x) after fdlibm's e_exp.c: x = k ln2 + r, |r| <= ln2/2; exp(r) = 1 - ((lo - r c / (2 - c)) - hi) with
// c = r - r^2 (P1 + r^2 (P2 + ...)); scaled by 2^k through the exponent field.  < 1 ulp.  Plain * and + only.
MUSE_HD double muse_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    const int k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double dk = (double)k;
    const double hi = x - dk * ln2HI, lo = dk * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    // y in [0.70, 1.42]; y * 2^k in one exact step where 2^k is a normal number, in two otherwise
    union { uint64_t u; double d; } s;
    if (k >= -1021 && k <= 1023) {
        s.u = (uint64_t)(k + 1023) << 52;
        return y * s.d;
    }
    if (k > 1023) {
        s.u = (uint64_t)(k - 1 + 1023) << 52;
        return (y * 2.0) * s.d;
    }
    s.u = (uint64_t)(k + 1000 + 1023) << 52;  // k >= -1075: a normal number
    return (y * s.d) * 9.33263618503218878990e-302;  // 2^-1000
}

// The block of element i when ntheta > kMaxTheta (the big tier of args.hpp, BigTheta): block k = floor(i * B / N) by arithmetic -- the same blocks as the boundaries
// bnd[k] = ceil(k N / B) of the small tiers' compare chain (models.hpp, block_of) -- instead of up to 63 compares.  i * B < 2^34 is exact in a double, the
// product with the rounded reciprocal is off by at most one, which the exact remainder corrects; the pad element of an
// odd-length vector and the phantom slots behind it (i >= N) belong to the last block, as in block_of.
MUSE_HD int block_of_big(int N, int B, double rcpN, int i) {
    const long long ab = (long long)i * B;
    int q = (int)((double)ab * rcpN);
    const long long r = ab - (long long)q * N;
    q += r >= N ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q < B - 1 ? q : B - 1;
}

// (component k's entries -- on the device one lane per component -- and the constant term, a sum in component order)
MUSE_HD void make_map_theta_component(int k, int ntheta, const double* theta, MapTheta& m) {
    const bool live = k < ntheta;
    m.t.theta[k] = live ? theta[k] : 0.0;
    m.t.sd[k] = live ? muse_exp(0.5 * theta[k]) : 0.0;
    m.t.iv[k] = live ? muse_exp(-theta[k]) : 0.0;
}
MUSE_HD void make_map_theta_const(int ntheta, const int64_t* bnd, const double* theta, MapTheta& m) {
    double cst = 0.0;
    for (int k = 0; k < ntheta; ++k) cst += (double)(bnd[k + 1] - bnd[k]) * theta[k];
    m.f_const = cst;
    m.pad_ = 0.0;
}
MUSE_HD void make_map_theta(int ntheta, const int64_t* bnd, const double* theta, MapTheta& m) {
    for (int k = 0; k < kMaxTheta; ++k) make_map_theta_component(k, ntheta, theta, m);
    make_map_theta_const(ntheta, bnd, theta, m);
}

// Work space of one step.  The caller places it: on the device in LDS -- a lane's local arrays indexed at run time live in
// scratch memory (HBM), and the dependent chains of a one-lane Gauss-Jordan then pay a memory round trip per access
// (measured: 50 us per step that way) -- on the host on the stack.
struct StepWork {
    double M[kMaxTheta][2 * kMaxTheta];
    double gprior[kMaxTheta], hprior[kMaxTheta];
    double Hlike[kMaxTheta * kMaxTheta], Hinv_like_inv[kMaxTheta * kMaxTheta], Hpost[kMaxTheta * kMaxTheta];
    double rec[7 * kMaxTheta + kMaxTheta * kMaxTheta + 1];
    double theta[kMaxTheta], theta_next[kMaxTheta + 1];   // (+1: the loop kernel's stepper appends its status word)
};

// inverse of a small dense matrix (n <= kMaxTheta) by Gauss-Jordan with partial pivoting; false if singular
MUSE_HD bool small_inverse(int n, const double* A, double* inv, double (*M)[2 * kMaxTheta]) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            M[i][j] = A[i * n + j];
            M[i][n + j] = i == j ? 1.0 : 0.0;
        }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r)
            if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (!(fabs(M[piv][col]) > 0.0)) return false;
        if (piv != col)
            for (int j = 0; j < 2 * n; ++j) {
                const double tmp = M[piv][j];
                M[piv][j] = M[col][j];
                M[col][j] = tmp;
            }
        const double d = M[col][col];
        for (int j = 0; j < 2 * n; ++j) M[col][j] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            const double f = M[r][col];
            if (f != 0.0)
                for (int j = 0; j < 2 * n; ++j) M[r][j] -= f * M[col][j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) inv[i * n + j] = M[i][n + j];
    return true;
}

enum { STEP_OK = 0, STEP_SINGULAR_LIKE = 1, STEP_SINGULAR_POST = 2, STEP_DOMAIN = 3 };

// mean and corrected variance of component k of the S simulation scores gs[s * ntheta + k] (src/muse.jl:183,188).
// Summation order (ONE shape for the host loop and for the device-resident loop, which runs it on a 64-lane wavefront;
// Julia's own sum is pairwise over blocks with a reassociating inner loop, so no order is "the reference's"): 64 partial
// sums, partial l over the simulations s = l, l + 64, ... in increasing s, then the balanced pairwise tree over l in natural
// order -- what reduce.hpp's wave_total computes with its four DPP steps and the final (r0 + r1) + (r2 + r3).
#if defined(__HIPCC__)
// (device form: called by ALL 64 lanes of one wavefront; every lane returns the same values)
__device__ inline void step_moments_wave(int lane, int k, int ntheta, int S, const double* gs, double& mean, double& var) {
    double m = 0.0;
    for (int s = lane; s < S; s += 64) m += gs[(int64_t)s * ntheta + k];
    m = wave_total<false>(m);
    m /= S;
    double v = 0.0;
    for (int s = lane; s < S; s += 64) {
        const double dlt = gs[(int64_t)s * ntheta + k] - m;
        v += dlt * dlt;
    }
    v = wave_total<false>(v);
    v /= (S - 1);  // corrected (src/muse.jl:188)
    mean = m;
    var = v;
}
#endif
inline double step_tree64(double* p) {   // in place: the balanced pairwise tree over 64 leaves in natural order
    for (int stride = 1; stride < 64; stride *= 2)
        for (int i = 0; i < 64; i += 2 * stride) p[i] = p[i] + p[i + stride];
    return p[0];
}
inline void step_moments(int k, int ntheta, int S, const double* gs, double& mean, double& var) {
    double part[64];
    for (int l = 0; l < 64; ++l) {
        double m = 0.0;
        for (int s = l; s < S; s += 64) m += gs[(int64_t)s * ntheta + k];
        part[l] = m;
    }
    double m = step_tree64(part);
    m /= S;
    for (int l = 0; l < 64; ++l) {
        double v = 0.0;
        for (int s = l; s < S; s += 64) {
            const double dlt = gs[(int64_t)s * ntheta + k] - m;
            v += dlt * dlt;
        }
        part[l] = v;
    }
    double v = step_tree64(part);
    v /= (S - 1);  // corrected (src/muse.jl:188)
    mean = m;
    var = v;
}

// One record of the history and the next iterate, from theta (where the map ran), the data element's score g_dat and the
// moments of the simulation scores: h = [theta, g_like_dat, g_like, g_prior, g_post, diag H^-1_like, diag H_prior]
// (ntheta each), H^-1_post (ntheta x ntheta); theta_next = theta - alpha H^-1_post g_post (src/muse.jl:183-208,224).
// In pieces per component, so that the device-resident loop can give every component a lane of its own (the pieces of one
// component touch nothing another component's pieces write, except where a barrier is noted); step_record is the pieces in
// component order -- the host loop's form, and the definition.
MUSE_HD void step_component(const StepParams& sp, int k, const double* theta, const double* g_dat, const double* mean, const double* var,
                            double* h, StepWork& w) {
    const int nt = sp.ntheta;
    if (sp.prior_kind == 1) {
        const double sg2 = sp.prior_sigma[k] * sp.prior_sigma[k];
        w.gprior[k] = -(theta[k] - sp.prior_mean[k]) / sg2;
        w.hprior[k] = -1.0 / sg2;
    } else {
        double zero = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(zero));   // (the loop kernel: else the 64-bit zero is formed before the stepper's loop and held -- in scratch --
                                         //  across the stepper's own solves; the same bits either way)
#endif
        w.gprior[k] = zero;
        w.hprior[k] = zero;
    }
    h[k] = theta[k];
    h[nt + k] = g_dat[k];                         // g_like_dat
    h[2 * nt + k] = g_dat[k] - mean[k];           // g_like  = g_dat - mean(g_sims)
    h[3 * nt + k] = w.gprior[k];
    h[4 * nt + k] = h[2 * nt + k] + w.gprior[k];  // g_post
    h[5 * nt + k] = -1.0 / var[k];                // diag H^-1_like
    h[6 * nt + k] = w.hprior[k];
}
// H^-1_post = inv(inv(H^-1_like) + H_prior) (src/muse.jl:208).  Both matrices are DIAGONAL here (the "sims" update and an
// independent prior), and for a diagonal matrix the dense inverse (small_inverse above: Gauss-Jordan with partial pivoting,
// kept as the definition and checked against this form bit for bit in tests/native/step_driver.cpp) does exactly this:
// no row is ever exchanged or eliminated (every off-diagonal entry is a zero), pivot i fails iff !(|d_i| > 0), the
// diagonal of the inverse is ONE division 1 / d_i, and the off-diagonal entries of row i are (+0) / d_i -- zeros that
// carry d_i's sign.  Written out, the step is 2 nt divisions instead of two eliminations whose every access, on the
// one lane of the device-resident loop that took the step, was a dependent LDS round trip (measured at nt = 4: 25 us).
MUSE_HD bool step_like_ok(int nt, int k, const double* h) { return fabs(h[5 * nt + k]) > 0.0; }
MUSE_HD bool step_post_diag(int nt, int k, const double* h, StepWork& w) {   // the diagonal of inv(H^-1_like) + H_prior; false: singular
    w.Hinv_like_inv[k] = 1.0 / h[5 * nt + k] + w.hprior[k];
    return fabs(w.Hinv_like_inv[k]) > 0.0;
}
// row a of H^-1_post and component a of the Newton-Raphson step (src/muse.jl:224); reads every component's g_post
MUSE_HD void step_row(const StepParams& sp, int a_, double* h, double* theta_next, StepWork& w) {
    const int nt = sp.ntheta;
    const double dg = 1.0 / w.Hinv_like_inv[a_], off = 0.0 / w.Hinv_like_inv[a_];
    for (int b = 0; b < nt; ++b) {
        w.Hpost[a_ * nt + b] = a_ == b ? dg : off;
        h[7 * nt + a_ * nt + b] = a_ == b ? dg : off;
    }
    double stp = 0.0;
    for (int b = 0; b < nt; ++b) stp += w.Hpost[a_ * nt + b] * h[4 * nt + b];
    theta_next[a_] = h[a_] - sp.alpha * stp;
}
MUSE_HD int step_record(const StepParams& sp, const double* theta, const double* g_dat, const double* mean, const double* var,
                        double* h, double* theta_next, StepWork& w) {
    const int nt = sp.ntheta;
    for (int k = 0; k < nt; ++k) step_component(sp, k, theta, g_dat, mean, var, h, w);
    for (int k = 0; k < nt; ++k)
        if (!step_like_ok(nt, k, h)) return STEP_SINGULAR_LIKE;
    bool post_ok = true;
    for (int k = 0; k < nt; ++k) post_ok = step_post_diag(nt, k, h, w) && post_ok;
    if (!post_ok) return STEP_SINGULAR_POST;
    for (int a_ = 0; a_ < nt; ++a_) step_row(sp, a_, h, theta_next, w);
    return STEP_OK;
}

// The convergence test at the top of an iteration i > 2 on the last two records h1 (newer) and h0 (src/muse.jl:163-166):
// 1 converged, 0 go on, -1 DomainError (sqrt of a negative number: H^-1_post' is not negative definite).  The quadratic
// form is a sum over the components of step_converged_term, in component order.
MUSE_HD double step_converged_term(int nt, int a_, const double* h1, const double* h0) {
    const double* Hp = h1 + 7 * nt;
    double row = 0.0;
    for (int b = 0; b < nt; ++b) row += Hp[a_ * nt + b] * (h1[b] - h0[b]);
    return (h1[a_] - h0[a_]) * row;
}
MUSE_HD int step_converged_from(double q, double theta_rtol) {
    if (-q < 0.0) return -1;  // a NaN compares false and the loop goes on
    return sqrt(-q) < theta_rtol ? 1 : 0;
}
MUSE_HD int step_converged(int nt, const double* h1, const double* h0, double theta_rtol) {
    double q = 0.0;
    for (int a_ = 0; a_ < nt; ++a_) q += step_converged_term(nt, a_, h1, h0);
    return step_converged_from(q, theta_rtol);
}

}  // namespace muse
