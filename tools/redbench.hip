// Cost of the solver's workgroup reduction (block_allreduce of csrc/reduce.hpp)
// at the solver's occupancy: 512 threads per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
namespace muse {
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }

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
    if constexpr (IS_MAX) return nanmax(a, b);
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

template <int T, int KS, int KM>
__device__ __forceinline__ void block_allreduce(double (&s)[KS > 0 ? KS : 1], double (&m)[KM > 0 ? KM : 1],
                                                double* red, int& parity, int tid) {
    constexpr int NW = T / 64, K = KS + KM;
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


}
using namespace muse;
#define REP 64
template <int MODE>
__global__ void __launch_bounds__(512) k(double* out, uint64_t* cyc, double a0) {
    __shared__ double red[2 * 8 * 8];
    const int tid = threadIdx.x;
    int parity = 0;
    double s[2] = {a0 + tid * 1e-3, a0 - tid * 1e-4}, m[1] = {a0 * tid};
    double acc = 0;
    __syncthreads();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        if (MODE == 0) {
            block_allreduce<512, 2, 1>(s, m, red, parity, tid);
        } else if (MODE == 1) {  // wave totals only
            s[0] = wave_total<false>(s[0]); s[1] = wave_total<false>(s[1]); m[0] = wave_total<true>(m[0]);
        } else if (MODE == 2) {  // one sum only, full block
            double z[1] = {0.0};
            double s1[1] = {s[0]};
            block_allreduce<512, 1, 0>(s1, z, red, parity, tid);
            s[0] = s1[0];
        } else if (MODE == 3) {  // barrier only
            __syncthreads();
        }
        acc += s[0] + s[1] + m[0];
        s[0] = s[0] * 1e-3 + tid; s[1] = s[1] * 1e-3 - tid; m[0] = m[0] * 1e-3 + tid * 0.5;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name) {
    double* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mm = 0; for (int i = 0; i < 256; ++i) mm += h[i]; mm /= 256;
    printf("%-34s %8.0f cycles per call\n", name, mm / REP);
}
int main() { run<0>("block_allreduce<512,2,1>"); run<1>("3 wave totals"); run<2>("block_allreduce<512,1,0>"); run<3>("barrier only"); return 0; }
