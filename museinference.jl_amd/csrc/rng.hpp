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

// ---- K chains side by side ---------------------------------------------------------------------------------
// At the two waves per SIMD the solver runs at, a wave issues a VALU instruction every ~5.3 cycles when it has four
// independent ones to choose from and every ~11 cycles when each depends on the one before (tools/clockprobe.hip).
// The generator is long dependent chains (Philox rounds, Horner polynomials, Newton steps), and the compiler keeps
// the chains of different elements one after the other, so every function below is written for K elements at once,
// one source statement = K independent instructions next to each other.  The arithmetic of each element is the
// sequence it always was (K = 1 is the scalar definition the oracle restates).
#define MUSE_K for (int k = 0; k < K; ++k)

// IEEE division a / b for operands in the normal range with a normal quotient (here b in [1.29, 2.42], |a| < 0.42):
// v_rcp_f64 + two Newton-Raphson steps + one correction, i.e. the sequence the compiler emits for `/` without the
// operand rescaling (v_div_scale), the scaled fma (v_div_fmas, which goes through VCC and so cannot interleave with
// a neighbour's) and the special-case fix-up (v_div_fixup) that only arguments outside that range need.
template <int K>
__device__ __forceinline__ void div_normal(const double (&a)[K], const double (&b)[K], double (&q)[K]) {
    double r[K], e[K], t[K];
#pragma unroll
    MUSE_K r[k] = __builtin_amdgcn_rcp(b[k]);
#pragma unroll
    MUSE_K e[k] = fma(-b[k], r[k], 1.0);
#pragma unroll
    MUSE_K r[k] = fma(r[k], e[k], r[k]);
#pragma unroll
    MUSE_K e[k] = fma(-b[k], r[k], 1.0);
#pragma unroll
    MUSE_K r[k] = fma(r[k], e[k], r[k]);
#pragma unroll
    MUSE_K t[k] = a[k] * r[k];
#pragma unroll
    MUSE_K e[k] = fma(-b[k], t[k], a[k]);
#pragma unroll
    MUSE_K q[k] = fma(e[k], r[k], t[k]);
}

// log(x) for x in (0,1): fdlibm-style reduction x = 2^k (1+f), log(1+f) by the degree-14
// odd series in s = f/(2+f).  Inputs here are never subnormal (x >= 2^-53).
template <int K>
__device__ __forceinline__ void log_unit(const double (&x)[K], double (&out)[K]) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double f[K], dk[K], hfsq[K], den[K], s[K], z[K], w[K], t1[K], t2[K], R[K];
#pragma unroll
    MUSE_K {
        uint64_t bits = (uint64_t)__double_as_longlong(x[k]);
        uint32_t hx = (uint32_t)(bits >> 32);
        hx += 0x3ff00000u - 0x3fe6a09eu;
        dk[k] = (double)((int)(hx >> 20) - 0x3ff);
        hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
        bits = ((uint64_t)hx << 32) | (bits & 0xffffffffull);
        f[k] = __longlong_as_double((long long)bits) - 1.0;
    }
#pragma unroll
    MUSE_K den[k] = 2.0 + f[k];
#pragma unroll
    MUSE_K hfsq[k] = 0.5 * f[k] * f[k];
    div_normal<K>(f, den, s);
#pragma unroll
    MUSE_K z[k] = s[k] * s[k];
#pragma unroll
    MUSE_K w[k] = z[k] * z[k];
#pragma unroll
    MUSE_K t1[k] = fma(w[k], Lg6, Lg4);
#pragma unroll
    MUSE_K t2[k] = fma(w[k], Lg7, Lg5);
#pragma unroll
    MUSE_K t1[k] = fma(w[k], t1[k], Lg2);
#pragma unroll
    MUSE_K t2[k] = fma(w[k], t2[k], Lg3);
#pragma unroll
    MUSE_K t1[k] = w[k] * t1[k];
#pragma unroll
    MUSE_K t2[k] = fma(w[k], t2[k], Lg1);
#pragma unroll
    MUSE_K t2[k] = z[k] * t2[k];
#pragma unroll
    MUSE_K R[k] = t2[k] + t1[k];
#pragma unroll
    MUSE_K out[k] = fma(dk[k], ln2_hi, (fma(s[k], hfsq[k] + R[k], dk[k] * ln2_lo) - hfsq[k]) + f[k]);
}

// sin(pi t), cos(pi t), t in [0,2): exact reduction to |r| <= 1/4, minimax kernels on pi r.
template <int K>
__device__ __forceinline__ void sincospi_02(const double (&t)[K], double (&sn)[K], double (&cs)[K]) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
                 C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11,
                 PI = 3.14159265358979311600e+00;
    int n[K];
    double y[K], z[K], w[K], sa[K], sb[K], ca[K], cb[K], v[K], ks[K], kc[K], hz[K], ww[K];
#pragma unroll
    MUSE_K n[k] = (int)(2.0 * t[k] + 0.5);
#pragma unroll
    MUSE_K y[k] = (t[k] - 0.5 * (double)n[k]) * PI;
#pragma unroll
    MUSE_K z[k] = y[k] * y[k];
#pragma unroll
    MUSE_K w[k] = z[k] * z[k];
    // rs = fma(z w, fma(z, S6, S5), fma(z, fma(z, S4, S3), S2));  rc = fma(w w, fma(z, fma(z, C6, C5), C4), z fma(z, fma(z, C3, C2), C1))
#pragma unroll
    MUSE_K sa[k] = fma(z[k], S6, S5);
#pragma unroll
    MUSE_K sb[k] = fma(z[k], S4, S3);
#pragma unroll
    MUSE_K ca[k] = fma(z[k], C6, C5);
#pragma unroll
    MUSE_K cb[k] = fma(z[k], C3, C2);
#pragma unroll
    MUSE_K sb[k] = fma(z[k], sb[k], S2);
#pragma unroll
    MUSE_K ca[k] = fma(z[k], ca[k], C4);
#pragma unroll
    MUSE_K cb[k] = fma(z[k], cb[k], C1);
#pragma unroll
    MUSE_K sa[k] = fma(z[k] * w[k], sa[k], sb[k]);          // rs
#pragma unroll
    MUSE_K ca[k] = fma(w[k] * w[k], ca[k], z[k] * cb[k]);   // rc
#pragma unroll
    MUSE_K v[k] = z[k] * y[k];
#pragma unroll
    MUSE_K hz[k] = 0.5 * z[k];
#pragma unroll
    MUSE_K ks[k] = fma(v[k], fma(z[k], sa[k], S1), y[k]);
#pragma unroll
    MUSE_K ww[k] = 1.0 - hz[k];
#pragma unroll
    MUSE_K kc[k] = ww[k] + fma(z[k], ca[k], (1.0 - ww[k]) - hz[k]);
#pragma unroll
    MUSE_K {
        const bool swap = (n[k] & 1) != 0;
        const double a = swap ? kc[k] : ks[k];  // |sin|
        const double b = swap ? ks[k] : kc[k];  // |cos|
        sn[k] = (n[k] & 2) ? -a : a;
        cs[k] = ((n[k] + 1) & 2) ? -b : b;
    }
}

// IEEE sqrt for an argument in the normal range (here 2.2e-16 <= x <= 73): the correctly rounding
// v_rsq_f64 + two Newton-Raphson steps that the compiler emits for sqrt(), without the rescaling and the
// zero/infinity selects that only arguments outside that range need.
template <int K>
__device__ __forceinline__ void sqrt_normal(const double (&x)[K], double (&out)[K]) {
    double y[K], s[K], h[K], r[K], d[K];
#pragma unroll
    MUSE_K y[k] = __builtin_amdgcn_rsq(x[k]);
#pragma unroll
    MUSE_K s[k] = x[k] * y[k];
#pragma unroll
    MUSE_K h[k] = 0.5 * y[k];
#pragma unroll
    MUSE_K r[k] = fma(-h[k], s[k], 0.5);
#pragma unroll
    MUSE_K s[k] = fma(s[k], r[k], s[k]);
#pragma unroll
    MUSE_K h[k] = fma(h[k], r[k], h[k]);
#pragma unroll
    MUSE_K d[k] = fma(-s[k], s[k], x[k]);
#pragma unroll
    MUSE_K s[k] = fma(d[k], h[k], s[k]);
#pragma unroll
    MUSE_K d[k] = fma(-s[k], s[k], x[k]);
#pragma unroll
    MUSE_K out[k] = fma(d[k], h[k], s[k]);
}

// Philox4x32-10 for K counters (i[k], sim) under one key: round r of every chain before round r + 1 of any.
template <int K>
__device__ __forceinline__ void philox4x32_10(const uint64_t (&i)[K], uint64_t sim, uint64_t seed, uint32_t (&out)[K][4]) {
    uint32_t c0[K], c1[K], c2[K], c3[K];
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    MUSE_K {
        c0[k] = (uint32_t)i[k];
        c1[k] = (uint32_t)(i[k] >> 32);
        c2[k] = (uint32_t)sim;
        c3[k] = (uint32_t)(sim >> 32);
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0[K], p1[K];
#pragma unroll
        MUSE_K p0[k] = (uint64_t)0xD2511F53u * c0[k];  // one v_mad_u64_u32 each (hi and lo together)
#pragma unroll
        MUSE_K p1[k] = (uint64_t)0xCD9E8D57u * c2[k];
#pragma unroll
        MUSE_K {
            const uint32_t n0 = (uint32_t)(p1[k] >> 32) ^ c1[k] ^ k0, n2 = (uint32_t)(p0[k] >> 32) ^ c3[k] ^ k1;
            c1[k] = (uint32_t)p1[k];
            c3[k] = (uint32_t)p0[k];
            c0[k] = n0;
            c2[k] = n2;
        }
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    MUSE_K { out[k][0] = c0[k]; out[k][1] = c1[k]; out[k][2] = c2[k]; out[k][3] = c3[k]; }
}

// The two standard normals of elements i[0..K) of simulation `sim`.
template <int K>
__device__ __forceinline__ void normal_pairs(uint64_t seed, uint64_t sim, const uint64_t (&i)[K], NormalPair (&np)[K]) {
    uint32_t w[K][4];
    philox4x32_10<K>(i, sim, seed, w);
    // u = (k + 1/2) 2^-52 with k the 52 random bits: put k in the mantissa of a double in [1,2),
    // subtract 1 (exact) and add 2^-53 (exact: (2k+1) 2^-53 has 53 significant bits).  No int->fp
    // conversion instructions; the value is identical to ((double)k + 0.5) * 2^-52.
    // The 64-bit pattern 0x3FF0000000000000 | (w_a << 20) | (w_b >> 12), one v_alignbit_b32 per half:
    //   low word  = ((w_a:w_b) >> 12)[31:0],  high word = ((0x3FF:w_a) >> 12)[31:0] = 0x3FF00000 | (w_a >> 12).
    double u1[K], t2[K], lg[K], r[K], sn[K], cs[K];
#pragma unroll
    MUSE_K {
        const double m1 = __hiloint2double((int)__builtin_amdgcn_alignbit(0x3FFu, w[k][0], 12),
                                           (int)__builtin_amdgcn_alignbit(w[k][0], w[k][1], 12));
        const double m2 = __hiloint2double((int)__builtin_amdgcn_alignbit(0x3FFu, w[k][2], 12),
                                           (int)__builtin_amdgcn_alignbit(w[k][2], w[k][3], 12));
        u1[k] = (m1 - 1.0) + 1.1102230246251565404e-16;
        t2[k] = 2.0 * ((m2 - 1.0) + 1.1102230246251565404e-16);
    }
    log_unit<K>(u1, lg);
#pragma unroll
    MUSE_K lg[k] = -2.0 * lg[k];
    sqrt_normal<K>(lg, r);
    sincospi_02<K>(t2, sn, cs);
#pragma unroll
    MUSE_K np[k] = NormalPair{r[k] * cs[k], r[k] * sn[k]};
}

__device__ __forceinline__ NormalPair normal_pair(uint64_t seed, uint64_t sim, uint64_t i) {
    const uint64_t ii[1] = {i};
    NormalPair np[1];
    normal_pairs<1>(seed, sim, ii, np);
    return np[0];
}
#undef MUSE_K

}  // namespace muse
