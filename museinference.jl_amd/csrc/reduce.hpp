// reduce.hpp -- workgroup-uniform scalars and the fixed-shape workgroup all-reduce (see muse_kernels.hip).
#pragma once
#include "args.hpp"

namespace muse {

// Maxima are taken with v_max_f64 (one instruction; a NaN operand is DROPPED), on |values| that start from 0.
// NaN propagation -- Julia's maximum(abs, g) is NaN if any element is -- comes from the sums reduced in the same
// pass: a NaN element makes the pass's objective / directional-derivative sum NaN, and the caller then replaces
// the maximum by NaN (nan_if).  (The explicit compare/select form cost 6 instructions per element and per
// reduction step.)
__device__ __forceinline__ double absmax(double a, double b) { return __builtin_fmax(a, __builtin_fabs(b)); }
__device__ __forceinline__ double nan_if(bool c, double v) { return c ? __builtin_nan("") : v; }

// Tell the compiler a value is workgroup-uniform (it is: every lane holds the same bits).  Control
// flow that depends on it then compiles to scalar branches and its live state to SGPRs.
__device__ __forceinline__ double uniform(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- fixed-shape all-reduce over the workgroup: KS sums and KM NaN-propagating maxima ------------
// Within a wave: four DPP exchange steps (xor 1, xor 2, half-row mirror, row mirror -- full-rate VALU
// moves, no LDS crossbar) leave each 16-lane row's total in all of its lanes; the four row totals are
// read with v_readlane into SGPRs and combined in a fixed order, so the wave total is a scalar.
// Across waves: lane 0 of each wave stores its total to LDS, ONE barrier, then lane l of every wave
// reads wave (l mod NW)'s total and a DPP butterfly over NW lanes + v_readfirstlane leaves the
// workgroup total in SGPRs of every wave.  The tree is the same for every thread, launch and GPU.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7-i within 8
constexpr int kDppMirror = 0x140;      // row_mirror:      lane i <-> 15-i within 16

template <bool IS_MAX>
__device__ __forceinline__ double combine(double a, double b) {
    if constexpr (IS_MAX) return __builtin_fmax(a, b);
    else return a + b;
}
template <bool IS_MAX>
__device__ __forceinline__ double wave_total(double v) {
    v = combine<IS_MAX>(v, dpp_move<kDppXor1>(v));
    v = combine<IS_MAX>(v, dpp_move<kDppXor2>(v));
    v = combine<IS_MAX>(v, dpp_move<kDppHalfMirror>(v));
    v = combine<IS_MAX>(v, dpp_move<kDppMirror>(v));
    const double r0 = read_lane(v, 0), r1 = read_lane(v, 16), r2 = read_lane(v, 32), r3 = read_lane(v, 48);
    return combine<IS_MAX>(combine<IS_MAX>(r0, r1), combine<IS_MAX>(r2, r3));
}
template <bool IS_MAX, int NW>
__device__ __forceinline__ double lanes_total(double v) {  // butterfly over the first NW (<= 16) lanes of a row
    if constexpr (NW >= 2) v = combine<IS_MAX>(v, dpp_move<kDppXor1>(v));
    if constexpr (NW >= 4) v = combine<IS_MAX>(v, dpp_move<kDppXor2>(v));
    if constexpr (NW >= 8) v = combine<IS_MAX>(v, dpp_move<kDppHalfMirror>(v));
    if constexpr (NW >= 16) v = combine<IS_MAX>(v, dpp_move<kDppMirror>(v));
    return uniform(v);
}

// RAW: the barrier alone (s_waitcnt lgkmcnt(0) + s_barrier) instead of __syncthreads(), whose fence also drains the
// vector-memory counter -- every store of the pass (the MAP going out in a solve's last pass: 80 KB per workgroup) and
// every LDS-DMA prefetch in flight would have to complete before the reduction could (measured at configs[1]: 48.9 -> 36.6 us
// per 512-sim step).  For code whose threads only ever meet through LDS and re-read from memory only what they stored
// themselves (the elementwise models); the stencil model's passes read what other threads stored.
template <bool RAW>
__device__ __forceinline__ void wg_barrier() {
    if constexpr (RAW) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}
template <int T, int KS, int KM, bool RAW = false>
__device__ __forceinline__ void block_allreduce(double (&s)[KS > 0 ? KS : 1], double (&m)[KM > 0 ? KM : 1],
                                                double* red, int& parity, int tid) {
    constexpr int NW = T / 64, K = KS + KM;
    // (laundered: otherwise the LDS addresses derived from tid are computed once per kernel, live -- and, in the
    // register-bound placements, spilled to scratch -- across the persistent loop, and every reduction waits for a
    // scratch reload before its LDS write)
    asm volatile("" : "+v"(tid));
    static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8 || NW == 16, "workgroup must be 2^k waves");
    static_assert(K <= 8, "reduction scratch holds 8 values per wave");
#pragma unroll
    for (int k = 0; k < KS; ++k) s[k] = wave_total<false>(s[k]);
#pragma unroll
    for (int k = 0; k < KM; ++k) m[k] = wave_total<true>(m[k]);
    double* buf = red + parity * (NW * 8);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) buf[wave * K + k] = s[k];
#pragma unroll
        for (int k = 0; k < KM; ++k) buf[wave * K + KS + k] = m[k];
    }
    __syncthreads();
    const int src = (tid & (NW - 1)) * K;
#pragma unroll
    for (int k = 0; k < KS; ++k) s[k] = lanes_total<false, NW>(buf[src + k]);
#pragma unroll
    for (int k = 0; k < KM; ++k) m[k] = lanes_total<true, NW>(buf[src + KS + k]);
    parity ^= 1;
}


}  // namespace muse
