// The step of the muse! loop (museinference.jl_amd/csrc/step.hpp: step_record) against its DEFINITION with the dense inverses
// (small_inverse: Gauss-Jordan with partial pivoting, as src/muse.jl:208 inverts general matrices): the diagonal closed form
// the library runs must give the same BITS -- signed zeros, infinities, NaNs and error codes included -- on random and on
// awkward inputs.  Also checks the 64-leaf summation tree of step_moments against a direct evaluation of the same tree.
// Exit 0 = identical everywhere.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../museinference.jl_amd/csrc/step.hpp"

using namespace muse;

static int dense_record(const StepParams& sp, const double* theta, const double* g_dat, const double* mean, const double* var,
                        double* h, double* theta_next, StepWork& w) {
    const int nt = sp.ntheta;
    double *gprior = w.gprior, *hprior = w.hprior, *Hlike = w.Hlike, *Hinv = w.Hinv_like_inv, *Hpost = w.Hpost;
    for (int k = 0; k < nt; ++k) {
        if (sp.prior_kind == 1) {
            const double sg2 = sp.prior_sigma[k] * sp.prior_sigma[k];
            gprior[k] = -(theta[k] - sp.prior_mean[k]) / sg2;
            hprior[k] = -1.0 / sg2;
        } else {
            gprior[k] = 0.0;
            hprior[k] = 0.0;
        }
        h[k] = theta[k];
        h[nt + k] = g_dat[k];
        h[2 * nt + k] = g_dat[k] - mean[k];
        h[3 * nt + k] = gprior[k];
        h[4 * nt + k] = h[2 * nt + k] + gprior[k];
        h[5 * nt + k] = -1.0 / var[k];
        h[6 * nt + k] = hprior[k];
    }
    for (int a = 0; a < nt * nt; ++a) Hlike[a] = 0.0;
    for (int k = 0; k < nt; ++k) Hlike[k * nt + k] = h[5 * nt + k];
    if (!small_inverse(nt, Hlike, Hinv, w.M)) return STEP_SINGULAR_LIKE;
    for (int k = 0; k < nt; ++k) Hinv[k * nt + k] += hprior[k];
    if (!small_inverse(nt, Hinv, Hpost, w.M)) return STEP_SINGULAR_POST;
    for (int a = 0; a < nt * nt; ++a) h[7 * nt + a] = Hpost[a];
    for (int a = 0; a < nt; ++a) {
        double stp = 0.0;
        for (int b = 0; b < nt; ++b) stp += Hpost[a * nt + b] * h[4 * nt + b];
        theta_next[a] = h[a] - sp.alpha * stp;
    }
    return STEP_OK;
}

static unsigned long long rng_state = 88172645463325252ull;
static double rnd() {  // xorshift -> (0, 1)
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return ((double)(rng_state >> 11) + 0.5) / 9007199254740992.0;
}

// block_of_big (the big tier's block index: a double product with 1/N and an exact remainder correction) against the integer
// definition floor(i B / N), clamped to the last block for the pad / phantom elements i >= N: every i for small N, every
// boundary's neighbourhood and random elements for N up to 2^28 - 1 (the engine's bound for 32-bit element indices).
static long check_blocks(int& bad) {
    long n = 0;
    auto one = [&](int N, int B, long long i) {
        if (i < 0 || i > 0x7fffffff) return;
        const long long want = i * B / N;
        const int expect = (int)(want < B - 1 ? want : B - 1);
        const int got = block_of_big(N, B, 1.0 / (double)N, (int)i);
        n += 1;
        if (got != expect && bad++ < 10) printf("block_of_big(N=%d, B=%d, i=%lld) = %d, floor = %d\n", N, B, i, got, expect);
    };
    for (int N = 1; N <= 400; ++N)
        for (int B = 1; B <= (N < kBigTheta ? N : kBigTheta); ++B)
            for (int i = 0; i <= N + 2; ++i) one(N, B, i);
    unsigned long long lcg = 88172645463325252ull;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(lcg >> 33); };
    const int sizes[] = {10000, 9999, 65536, 70001, 100000, 1000000, 999983, 16777216, 16777217, 134217728, 268435455, 268435399, 200000003};
    for (int N : sizes)
        for (int B = 2; B <= kBigTheta; ++B) {
            for (int k = 0; k <= B; ++k) {
                const long long bnd = ((long long)k * N + B - 1) / B;   // ceil(k N / B): the first element of block k
                for (int d = -3; d <= 3; ++d) one(N, B, bnd + d);
            }
            for (int r = 0; r < 64; ++r) one(N, B, rnd() % (unsigned)N);
            one(N, B, N); one(N, B, (long long)N + 1);
        }
    for (int r = 0; r < 200000; ++r) {
        const int N = 1 + (int)(rnd() % 268435455u), B = 1 + (int)(rnd() % (unsigned)(N < kBigTheta ? N : kBigTheta));
        const int k = (int)(rnd() % (unsigned)(B + 1));
        const long long bnd = ((long long)k * N + B - 1) / B;
        for (int d = -2; d <= 2; ++d) one(N, B, bnd + d);
        one(N, B, rnd() % (unsigned)N);
    }
    return n;
}

int main() {
    const double special[] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 1e-320, -1e-320, 1e308, -1e308, 1.0, -1.0};
    int bad = 0;
    long cases = 0;
    const long nblocks = check_blocks(bad);
    printf("block_of_big: %ld elements checked\n", nblocks);
    for (int nt = 1; nt <= kMaxTheta; ++nt)
        for (int rep = 0; rep < 4000; ++rep) {
            StepParams sp;
            memset(&sp, 0, sizeof sp);
            sp.ntheta = nt;
            sp.nsims = 64;
            sp.prior_kind = rep & 1;
            sp.alpha = 0.7;
            double theta[kMaxTheta], gd[kMaxTheta], mean[kMaxTheta], var[kMaxTheta];
            for (int k = 0; k < nt; ++k) {
                sp.prior_mean[k] = rnd() - 0.5;
                sp.prior_sigma[k] = 0.1 + 5.0 * rnd();
                theta[k] = 4.0 * rnd() - 2.0;
                gd[k] = 1e3 * (rnd() - 0.5);
                mean[k] = 1e3 * (rnd() - 0.5);
                var[k] = 1e4 * rnd();
                if (rep % 7 == 3 && (int)(rnd() * nt) == k) var[k] = special[(int)(rnd() * 11)];      // awkward variances
                if (rep % 11 == 5 && (int)(rnd() * nt) == k) gd[k] = special[(int)(rnd() * 11)];      // ... and scores
                if (rep % 13 == 6 && (int)(rnd() * nt) == k) sp.prior_sigma[k] = special[(int)(rnd() * 11)];
                if (rep % 17 == 9) var[k] = -var[k];                                                  // H^-1_like of the wrong sign
            }
            StepWork w1, w2;
            memset(&w1, 0, sizeof w1);
            memset(&w2, 0, sizeof w2);
            double tn1[kMaxTheta + 1] = {0}, tn2[kMaxTheta + 1] = {0};
            const int e1 = step_record(sp, theta, gd, mean, var, w1.rec, tn1, w1), e2 = dense_record(sp, theta, gd, mean, var, w2.rec, tn2, w2);
            cases += 1;
            const size_t nrec = (size_t)(7 * nt + nt * nt) * sizeof(double);
            bool same = e1 == e2;
            if (same && e1 == STEP_OK) same = memcmp(w1.rec, w2.rec, nrec) == 0 && memcmp(tn1, tn2, nt * sizeof(double)) == 0;
            if (same && e1 != STEP_OK) same = memcmp(w1.rec, w2.rec, (size_t)(7 * nt) * sizeof(double)) == 0;   // (what was written before the error)
            if (!same && bad++ < 5) fprintf(stderr, "nt %d rep %d: error %d vs %d, or different bits\n", nt, rep, e1, e2);
        }
    // the moments' summation tree: 64 strided partial sums, then the balanced pairwise tree in natural order
    for (int S = 2; S <= 700; S += 37) {
        double gs[700 * 3];
        for (int i = 0; i < S * 3; ++i) gs[i] = 1e3 * (rnd() - 0.5);
        for (int k = 0; k < 3; ++k) {
            double m, v;
            step_moments(k, 3, S, gs, m, v);
            double part[64];
            for (int l = 0; l < 64; ++l) { part[l] = 0.0; for (int s = l; s < S; s += 64) part[l] += gs[s * 3 + k]; }
            double lvl[64];
            memcpy(lvl, part, sizeof lvl);
            for (int n = 64; n > 1; n /= 2) for (int i = 0; i < n / 2; ++i) lvl[i] = lvl[2 * i] + lvl[2 * i + 1];
            const double mm = lvl[0] / S;
            if (memcmp(&mm, &m, 8) != 0 && bad++ < 5) fprintf(stderr, "moments: S %d k %d\n", S, k);
            (void)v;
        }
    }
    if (bad) { fprintf(stderr, "%d mismatches in %ld cases\n", bad, cases); return 1; }
    printf("step driver ok: %ld cases\n", cases);
    return 0;
}
