// rng.hpp -- counter-based normal stream for the HIP engine (gfx950).
//
// Stream definition (build-defined; Julia's RNG streams are not reproducible without Julia,
// SURVEY.md §7 "RNG parity"): Philox4x32-10 (Salmon et al., SC'11) with
//   key = (seed lo32, seed hi32), counter = (i lo32, i hi32, sim lo32, sim hi32)
// one call per element i of simulation `sim`; the four output words give two uniforms on
// (0,1) with 52 random bits each, mapped to two standard normals by Box-Muller.  log and
// sin/cos(pi t) are fixed polynomial sequences in IEEE +,-,*,/,sqrt and explicit fma only (this
// translation unit is compiled with -ffp-contract=off: no implicit contraction), so a stream depends on (seed, sim, i) alone:
// the same on every GPU, for every launch geometry, and bit-equal to a host evaluation of
// the same sequence.  This replaces split_rng (reference src/util.jl:87-92): "stream =
// f(master rng, sim index), never advanced by the drivers".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace muse {

struct NormalPair {
    double n1, n2;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;  // one v_mad_u64_u32 each (hi and lo together)
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// log(x) for x in (0,1): fdlibm-style reduction x = 2^k (1+f), log(1+f) by the degree-14
// odd series in s = f/(2+f).  Inputs here are never subnormal (x >= 2^-53).
__device__ __forceinline__ double log_unit(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(bits >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    bits = ((uint64_t)hx << 32) | (bits & 0xffffffffull);
    const double m = __longlong_as_double((long long)bits);
    const double f = m - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double dk = (double)k;
    return fma(dk, ln2_hi, (fma(s, hfsq + R, dk * ln2_lo) - hfsq) + f);
}

// sin(pi t), cos(pi t), t in [0,2): exact reduction to |r| <= 1/4, minimax kernels on pi r.
__device__ __forceinline__ void sincospi_02(double t, double& sn, double& cs) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
                 C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11,
                 PI = 3.14159265358979311600e+00;
    const int n = (int)(2.0 * t + 0.5);
    const double r = t - 0.5 * (double)n;
    const double y = r * PI;
    const double z = y * y;
    const double w = z * z;
    const double rs = fma(z * w, fma(z, S6, S5), fma(z, fma(z, S4, S3), S2));
    const double v = z * y;
    const double ks = fma(v, fma(z, rs, S1), y);
    const double rc = fma(w * w, fma(z, fma(z, C6, C5), C4), z * fma(z, fma(z, C3, C2), C1));
    const double hz = 0.5 * z;
    const double ww = 1.0 - hz;
    const double kc = ww + fma(z, rc, (1.0 - ww) - hz);
    const bool swap = (n & 1) != 0;
    const double a = swap ? kc : ks;  // |sin|
    const double b = swap ? ks : kc;  // |cos|
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -b : b;
}

// IEEE sqrt for an argument in the normal range (here 2.2e-16 <= x <= 73): the correctly rounding
// v_rsq_f64 + two Newton-Raphson steps that the compiler emits for sqrt(), without the rescaling and the
// zero/infinity selects that only arguments outside that range need.
__device__ __forceinline__ double sqrt_normal(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y;
    const double h0 = 0.5 * y;
    const double r0 = fma(-h0, s0, 0.5);
    const double s1 = fma(s0, r0, s0);
    const double h1 = fma(h0, r0, h0);
    const double d0 = fma(-s1, s1, x);
    const double s2 = fma(d0, h1, s1);
    const double d1 = fma(-s2, s2, x);
    return fma(d1, h1, s2);
}

__device__ __forceinline__ NormalPair normal_pair(uint64_t seed, uint64_t sim, uint64_t i) {
    uint32_t w[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)sim, (uint32_t)(sim >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), w);
    // u = (k + 1/2) 2^-52 with k the 52 random bits: put k in the mantissa of a double in [1,2),
    // subtract 1 (exact) and add 2^-53 (exact: (2k+1) 2^-53 has 53 significant bits).  No int->fp
    // conversion instructions; the value is identical to ((double)k + 0.5) * 2^-52.
    // The 64-bit pattern 0x3FF0000000000000 | (w_a << 20) | (w_b >> 12), one v_alignbit_b32 per half:
    //   low word  = ((w_a:w_b) >> 12)[31:0],  high word = ((0x3FF:w_a) >> 12)[31:0] = 0x3FF00000 | (w_a >> 12).
    const double m1 = __hiloint2double((int)__builtin_amdgcn_alignbit(0x3FFu, w[0], 12),
                                       (int)__builtin_amdgcn_alignbit(w[0], w[1], 12));
    const double m2 = __hiloint2double((int)__builtin_amdgcn_alignbit(0x3FFu, w[2], 12),
                                       (int)__builtin_amdgcn_alignbit(w[2], w[3], 12));
    const double u1 = (m1 - 1.0) + 1.1102230246251565404e-16;
    const double u2 = (m2 - 1.0) + 1.1102230246251565404e-16;
    const double r = sqrt_normal(-2.0 * log_unit(u1));
    double sn, cs;
    sincospi_02(2.0 * u2, sn, cs);
    return NormalPair{r * cs, r * sn};
}

}  // namespace muse
