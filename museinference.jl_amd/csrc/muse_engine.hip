// muse_engine.hip -- MI355X (gfx950) engine for the MUSE inner loop: one persistent workgroup per
// Monte-Carlo element runs  sample_x_z -> L-BFGS/HagerZhang MAP over z -> grad_theta score
// entirely on the device, one launch per batch (reference: the pmap bodies of muse!/get_J!/get_H!,
// src/muse.jl:169-176, :508-525, :426-442; the solver behind zhat_at_theta, src/interface.jl:162-171).
//
// Layout of one element's work on the chip
//   * thread t of the workgroup owns the element pairs q = t + j*T  (elements 2q, 2q+1): 16 B per
//     lane, 1 KiB per wave-instruction, for every vector in HBM (x, z, g, s, trial gradient,
//     the 2*m history vectors) -- fully coalesced.
//   * two storage policies share ONE solver source (the Vec accessors below), so their results are
//     bitwise identical:
//       - Resident: z, s and the trial gradient live in registers, x and g in LDS; only the L-BFGS
//         history (dx, dg pairs) and the final zhat touch HBM.  N <= kMaxResidentN (LDS-bound).
//       - Streaming: every vector lives in the workgroup's HBM scratch; any N.
//   * all reductions are fixed-shape (per-thread sequential over j, 64-lane xor butterfly, then the
//     wave partials summed in wave order by every thread), so a result depends only on
//     (seed, sim, theta, N): not on the grid, the GPU count or the storage policy.
//   * no MFMA: this is elementwise + reduction work (0.3-0.5 flop/B); the bound is HBM/LDS traffic
//     and fp64 transcendental issue in the sampler.
//
// The scalar control logic (HagerZhang line search, L-BFGS bookkeeping) is evaluated redundantly
// and identically by every thread from the broadcast reduction results: no divergence, no
// single-thread serial sections, one barrier per reduction.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/muse_hip.h"
#include "rng.hpp"

namespace muse {

constexpr int kM = 10;         // L-BFGS memory (Optim.LBFGS default m)
constexpr int kMaxIter = 1000;  // Optim.Options default iterations
constexpr int kMaxTheta = MUSE_MAX_THETA;
constexpr int kResultAreas = 4;
constexpr int kMaxCluster = 16;
constexpr int64_t kClusterMinN = 65536;  // N >= this: several workgroups cooperate on one problem
constexpr int64_t kMaxResidentN = 10000;

// HagerZhang() defaults of LineSearches.jl
constexpr double kHzDelta = 0.1, kHzSigma = 0.9, kHzRho = 5.0, kHzEpsilon = 1e-6, kHzGamma = 0.66, kHzPsi3 = 0.1;
constexpr int kHzLinesearchMax = 50, kHzIterFiniteMax = 52;
constexpr double kEps = 2.220446049250313e-16;

struct ThetaSet {
    double theta[kMaxTheta];
    double sd[kMaxTheta];  // exp(theta/2), host libm
    double iv[kMaxTheta];  // exp(-theta),  host libm
};

enum { X_SAMPLE = 0, X_DATA = 1, X_GIVEN = 2 };
enum { Z0_ZERO = MUSE_Z0_ZERO, Z0_TRUE = MUSE_Z0_TRUE, Z0_WARM = MUSE_Z0_WARM, Z0_COPY = 3 };
enum { BATCH_STD = 0, BATCH_FD = 1, BATCH_SINGLE = 2, BATCH_IMPLICIT = 3 };

struct BatchArgs {
    int64_t N, ld;
    int ntheta, kind;
    int64_t bnd[kMaxTheta + 1];  // block k = elements [bnd[k], bnd[k+1])
    int bnd32[kMaxTheta + 1];    // the same, 32-bit (N < 2^28), for the per-element block lookup
    int pad0_;
    uint64_t seed;
    double atol, f_const;  // f_const = sum_k N_k theta_k (constant term of -2 logLike)
    int nproblems, include_data, z0_mode, store_zhat;
    int cg_maxiter;   // BATCH_IMPLICIT: IterativeSolvers.cg maxiter (reference default 100)
    int debug;        // profiling aids: bit0 skip the solve (sample + score only), bit1 take x from the data vector
    int64_t sim_begin, fid_slot, slot0;
    ThetaSet tmap;                 // theta of the MAP problem and of the score
    const ThetaSet* tsample;       // FD: [2*ntheta] sampling thetas (plus, minus per column); else null
    const double* x_data;          // [ld]
    const double* x_given;         // BATCH_SINGLE: [ld]
    double* zhat;                  // [slots][ld]
    double* scores;                // [nproblems][ntheta]
    muse_info* info;               // [nproblems]
    double* scratch;               // per workgroup
    int64_t scratch_stride;        // doubles per workgroup
    int* work_counter;             // monotonically increasing ticket counter (never reset)
    int ticket_base, pad2_;        // this launch's tickets are work_counter values base .. base+nproblems-1
    // cluster mode (several workgroups per problem): csize workgroups 0..csize-1 of cluster blockIdx/csize
    int csize, nclusters;
    unsigned int* cl_counter;      // [nclusters] arrival counters (zeroed per launch)
    double* cl_part;               // [nclusters][2][csize][8] partial sums / maxima
    int* error_flag;               // set when a bounded cluster wait expires
    unsigned long long* stamps;    // diagnostic build (-DMUSE_STAMPS) only: [nproblems][16] shader-clock stamps
    // Standard normals of simulation streams already drawn inside the SAME host call (muse_run's later
    // iterations re-draw every simulation at a new theta, the FD batch draws each simulation 2*ntheta times):
    // [slot][2][ld], slot = sim - ncache_sim0.  mode 1: generate and store; mode 2: load instead of generating.
    double* ncache;
    int64_t ncache_sim0;
    int ncache_count, ncache_mode;
    int nstd;  // BATCH_STD: elements >= nstd only draw (and store) the normals of sim norm_sim0 + (p - nstd)
    int pad3_;
    int64_t norm_sim0;
};

struct ProblemDesc {
    int64_t sim;
    int nslot;         // slot of the simulation's normals in the cache, -1: none
    bool normals_only;
    int x_mode, z0_mode, tsample;  // tsample < 0: sample at tmap
    int64_t zslot, z0slot;         // zslot < 0: zhat not stored
};

__device__ __forceinline__ ProblemDesc describe(const BatchArgs& a, int p) {
    ProblemDesc d;
    d.normals_only = false;
    if (a.kind == BATCH_STD && p >= a.nstd) {
        d.sim = a.norm_sim0 + (p - a.nstd);
        d.x_mode = X_SAMPLE;
        d.z0_mode = Z0_ZERO;
        d.tsample = -1;
        d.zslot = -1;
        d.z0slot = a.slot0;
        d.normals_only = true;
    } else if (a.kind == BATCH_STD) {
        const bool data = a.include_data && p == 0;
        d.sim = data ? -1 : a.sim_begin + p - (a.include_data ? 1 : 0);
        d.x_mode = (data || (a.debug & 2)) ? X_DATA : X_SAMPLE;
        d.z0_mode = (data && a.z0_mode == Z0_TRUE) ? Z0_ZERO : a.z0_mode;
        d.tsample = -1;
        d.zslot = a.store_zhat ? a.slot0 + p : -1;
        d.z0slot = a.slot0 + p;
    } else if (a.kind == BATCH_FD) {
        const int per = 2 * a.ntheta;
        d.sim = a.sim_begin + p / per;
        d.x_mode = X_SAMPLE;
        d.z0_mode = Z0_COPY;
        d.tsample = p % per;
        d.zslot = -1;
        d.z0slot = a.fid_slot >= 0 ? a.fid_slot : a.slot0 + p / per;
    } else if (a.kind == BATCH_IMPLICIT) {
        d.sim = a.sim_begin + p;
        d.x_mode = X_SAMPLE;
        d.z0_mode = Z0_ZERO;  // zhat_guess_from_truth = zero(z) (src/muse.jl:343, src/interface.jl:184-186)
        d.tsample = -1;
        d.zslot = -1;
        d.z0slot = a.slot0;
    } else {
        d.sim = -1;
        d.x_mode = X_GIVEN;
        d.z0_mode = Z0_WARM;
        d.tsample = -1;
        d.zslot = a.slot0;
        d.z0slot = a.slot0;
    }
    const int64_t ns = d.sim - a.ncache_sim0;
    d.nslot = (a.ncache && d.sim >= 0 && ns >= 0 && ns < a.ncache_count) ? (int)ns : -1;
    return d;
}

// ------------------------------------------------------------------------------------------------
// Vec accessors: (jj, i) = (register slot, element index).  Register vectors ignore i, memory
// vectors ignore jj; the solver source is written once against this interface.
//
// Every access is UNCONDITIONAL (no `if (i < N)` around it), so the compiler can issue all of a
// thread's loads of a pass back to back and expose the memory latency once, not once per element:
//   * a thread's slots beyond the vector are "phantom zeros": register slots are cleared, HBM vectors
//     sit behind range-checked buffer descriptors (out-of-range loads return 0, stores are dropped),
//     LDS vectors redirect out-of-range indices to a dummy slot that holds 0;
//   * vectors are padded to an even length ld >= N and the pad element is kept at 0;
//   * every model maps (x, z) = (0, 0) to a zero gradient / zero objective and score terms, so the
//     phantoms contribute exact zeros to every reduction and write zeros back.
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) double lds_double;
// words shared between workgroups inside a launch are accessed through GLOBAL (never flat) pointers
typedef __attribute__((address_space(1))) double gf64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) int gi32;

// Buffer descriptor over `bytes` bytes at `base` (both workgroup-uniform; the readfirstlanes make
// that provable so that no waterfall loop is generated around the buffer instructions).
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int64_t bytes) {
    const uint64_t b = (uint64_t)base;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

template <int NR>
struct RegVec {
    double r[NR > 0 ? NR : 1];
    __device__ __forceinline__ double get(int jj, int) const { return r[jj]; }
    __device__ __forceinline__ void set(int jj, int, double v) { r[jj] = v; }
    // Unconditional definition of every slot at the point where a problem first defines the vector:
    // otherwise the previous problem's values stay live across the persistent loop.
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < (NR > 0 ? NR : 1); ++j) r[j] = 0.0;
    }
    template <int UU>
    __device__ __forceinline__ void flush(int, int) {}
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int off8(int i) { return (int)((unsigned)i << 3); }  // byte offset of element i (may exceed 2^31: unsigned)
__device__ __forceinline__ double load_f64(const rsrc_t& rs, int i) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off8(i), 0, 0);
    return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}
__device__ __forceinline__ void load_f64x2(const rsrc_t& rs, int i, double& d0, double& d1) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off8(i), 0, 0);
    d0 = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
    d1 = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
}
__device__ __forceinline__ void store_f64x2(const rsrc_t& rs, int i, double d0, double d1) {
    const long long b0 = __double_as_longlong(d0), b1 = __double_as_longlong(d1);
    u32x4 v;
    v.x = (unsigned)(b0 & 0xffffffffll);
    v.y = (unsigned)(b0 >> 32);
    v.z = (unsigned)(b1 & 0xffffffffll);
    v.w = (unsigned)(b1 >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off8(i), 0, 0);
}
// A vector in HBM, resident policy (history pairs, zhat): the element loop visits a thread's two
// adjacent elements (jj even then jj odd) back to back, so the pair moves with ONE 16-byte buffer
// instruction: the load is issued at jj even and its second half served at jj odd; a store is staged at
// jj even and issued at jj odd (buffer_load/store_dwordx4, 1 KiB per wave-instruction).
struct BufVec2 {
    rsrc_t rsrc;
    mutable double c0, c1;
    __device__ __forceinline__ void bind(const double* base, int64_t ld) { rsrc = make_rsrc(base, ld * 8); }
    __device__ __forceinline__ double get(int jj, int i) const {
        if ((jj & 1) == 0) {
            load_f64x2(rsrc, i, c0, c1);
            return c0;
        }
        return c1;
    }
    __device__ __forceinline__ void set(int jj, int i, double d) {
        if ((jj & 1) == 0) c0 = d;
        else store_f64x2(rsrc, i - 1, c0, d);
    }
    __device__ __forceinline__ void clear() {}
    template <int UU>
    __device__ __forceinline__ void flush(int, int) {}
};
// A vector in HBM, streaming policy.  A streaming pass handles U of a thread's pairs per trip
// ("chunk", jj = 2u, 2u+1): gets are 16-byte loads issued where they appear, sets are STAGED in
// registers and written by flush() after the chunk's U element bodies have run.  Inside a chunk no
// store sits between the loads, so the compiler issues all of the chunk's loads back to back and a wave
// has U x (vectors read) 1-KiB requests in flight per trip -- with the stores interleaved (hipcc cannot
// prove the buffers distinct) every pair's loads waited for the previous pair's store to issue and a
// pass paid one memory round trip per pair.  get1() is an 8-byte load at an arbitrary element (stencil
// neighbours).
template <int U>
struct BufChunk {
    rsrc_t rsrc;
    mutable double c1[U];
    double st[2 * U];
    __device__ __forceinline__ void bind(const double* base, int64_t ld) { rsrc = make_rsrc(base, ld * 8); }
    __device__ __forceinline__ double get(int jj, int i) const {
        if ((jj & 1) == 0) {
            double d0;
            load_f64x2(rsrc, i, d0, c1[jj >> 1]);
            return d0;
        }
        return c1[jj >> 1];
    }
    __device__ __forceinline__ double get1(int i) const { return load_f64(rsrc, i); }
    __device__ __forceinline__ void set(int jj, int, double d) { st[jj] = d; }
    __device__ __forceinline__ void clear() {}
    // the UU (<= U) pairs i0, i0 + 2*pstride, ... of the trip that started at element i0
    template <int UU>
    __device__ __forceinline__ void flush(int i0, int pstride) {
        static_assert(UU <= U, "trip longer than the staging area");
#pragma unroll
        for (int u = 0; u < UU; ++u) store_f64x2(rsrc, i0 + 2 * u * pstride, st[2 * u], st[2 * u + 1]);
    }
};
// flush-list entry for a vector that a pass writes only under a (workgroup-uniform) condition
template <class V>
struct FlushIf {
    V& v;
    bool on;
    template <int UU>
    __device__ __forceinline__ void flush(int i0, int pstride) {
        if (on) v.template flush<UU>(i0, pstride);
    }
};
template <class V>
__device__ __forceinline__ FlushIf<V> when(bool on, V& v) { return FlushIf<V>{v, on}; }
// A vector in LDS; p[ld], p[ld+1] is a dummy pair that holds 0.  Like BufVec2 it moves a thread's two adjacent
// elements with one instruction (ds_read2_b64 / ds_write2_b64) and one address computation: the pair is read
// at jj even (second half served at jj odd), a store is staged at jj even and issued at jj odd.
struct LdsVec {
    lds_double* p;
    int ld;
    mutable double c1;
    double s0;
    __device__ __forceinline__ void bind(double* base, int64_t ld_) {
        p = (lds_double*)base;
        ld = (int)ld_;
    }
    __device__ __forceinline__ double get(int jj, int i) const {
        if ((jj & 1) == 0) {
            const lds_double* q = p + (i < ld ? i : ld);
            const double d0 = q[0];
            c1 = q[1];
            return d0;
        }
        return c1;
    }
    __device__ __forceinline__ void set(int jj, int i, double v) {
        if ((jj & 1) == 0) {
            s0 = v;
        } else {
            lds_double* q = p + (i - 1 < ld ? i - 1 : ld);
            q[0] = s0;
            q[1] = v;
        }
    }
    __device__ __forceinline__ void clear() {}
    template <int UU>
    __device__ __forceinline__ void flush(int, int) {}
};

// Element loop of one thread: pairs q = tid + j*T, elements 2q and 2q+1, in increasing j; no bounds
// checks (see above).  EPT > 0: compile-time trip count, fully unrolled (register slots are static).
// EPT == 0 (streaming): U pairs per trip with jj = 2u, 2u+1; the vectors the pass writes are listed
// after the body and flushed once per trip (BufChunk).  A trip may reach up to U-1 pairs past the end
// of the vector: those are phantom zeros like every other out-of-range slot.
// Element indices are 32-bit (N < 2^28), and tid is laundered through an empty asm so that the
// per-slot offsets are recomputed in each pass (two integer ops) instead of being hoisted out of
// the persistent loop and held -- or spilled -- for the kernel's lifetime.
template <int T, int EPT, int U, class F, class... W>
__device__ __forceinline__ void for_elems(int64_t ld, int tfirst, int pstride, F&& f, W&&... written) {
    int t = tfirst;
    asm volatile("" : "+v"(t));
    if constexpr (EPT > 0) {
        (void)pstride;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int i0 = 2 * (t + j * T);
            f(2 * j, i0);
            f(2 * j + 1, i0 + 1);
        }
    } else {
        // streaming: pairs tfirst, tfirst + pstride, ... (pstride = T, or csize*T in cluster mode)
        const int n = (int)ld;
#pragma unroll 1
        for (int i0 = 2 * t; i0 < n; i0 += 2 * U * pstride) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 2 * u * pstride;
                f(2 * u, i);
                f(2 * u + 1, i + 1);
            }
            (written.template flush<U>(i0, pstride), ...);
        }
    }
}

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

// ------------------------------------------------------------------------------------------------
// Models.  grad() returns d(-logLike)/dz_i and adds the element's share of -2 logLike (without the
// constant) to facc; the score is assembled from per-block sums of score_term().
template <int MAXB = kMaxTheta>
__device__ __forceinline__ int block_of(const BatchArgs& a, int i) {
    int k = 0;
#pragma unroll
    for (int b = 1; b < MAXB; ++b) k += (i >= a.bnd32[b]) ? 1 : 0;  // bnd32[b] = INT_MAX for b >= ntheta
    return k;
}

template <int MAXB_>
struct FunnelModel {  // z_i ~ N(0, e^theta_k), x_i ~ N(z_i, 1)
    static constexpr int MAXB = MAXB_;
    static constexpr bool kStencil = false;
    static constexpr int kId = MUSE_MODEL_FUNNEL;
    __device__ static __forceinline__ void sample(double sd, double n1, double n2, double& z, double& x) {
        z = sd * n1;
        x = z + n2;
    }
    __device__ static __forceinline__ double grad(double iv, double x, double z, double& facc) {
        const double r = x - z, t = iv * z;
        facc = fma(t, z, fma(r, r, facc));
        return t - r;
    }
    __device__ static __forceinline__ double score_term(double, double z) { return z * z; }
};
struct NoiseModel {  // z_i ~ N(0,1), x_i ~ N(z_i, e^theta)
    static constexpr int MAXB = 1;
    static constexpr bool kStencil = false;
    static constexpr int kId = MUSE_MODEL_NOISE;
    __device__ static __forceinline__ void sample(double sd, double n1, double n2, double& z, double& x) {
        z = n1;
        x = n1 + sd * n2;
    }
    __device__ static __forceinline__ double grad(double iv, double x, double z, double& facc) {
        const double r = x - z, t = iv * r;
        facc = fma(z, z, fma(t, r, facc));
        return z - t;
    }
    __device__ static __forceinline__ double score_term(double x, double z) {
        const double r = x - z;
        return r * r;
    }
};
template <int MAXB_>
struct SmoothModel {  // z as funnel, x = A z + n, A = periodic (1/4, 1/2, 1/4); streaming policy only
    static constexpr int MAXB = MAXB_;
    static constexpr bool kStencil = true;
    static constexpr int kId = MUSE_MODEL_SMOOTH;
    __device__ static __forceinline__ double score_term(double, double z) { return z * z; }
};

// ------------------------------------------------------------------------------------------------
// Storage policies.
template <int T_, bool CLUSTER = false, int U_ = 4>
struct PlaceStreaming {
    static constexpr int T = T_, EPT = 0, U = U_;  // U pairs of a thread per trip of a streaming pass
    // two waves per SIMD: 2 workgroups of 256 threads (cluster mode sizes its grid from that) or 1 of 512 per CU,
    // i.e. a budget of 256 registers per lane
    static constexpr int kWavesPerEu = 2;
    static constexpr bool kResident = false, kXgLds = false, kCluster = CLUSTER;
    using VX = BufChunk<U_>;
    using VG = VX; using VZ = VX; using VS = VX;
    using VH = VX;
};
template <int T_, int EPT_, bool XG_LDS>
struct PlaceResident {
    static constexpr int T = T_, EPT = EPT_, U = 1;
    static constexpr int kWavesPerEu = 1;  // no lower bound beyond the launch bounds
    static constexpr bool kResident = true, kXgLds = XG_LDS, kCluster = false;
    using VX = typename std::conditional<XG_LDS, LdsVec, RegVec<2 * EPT_>>::type;
    using VG = VX;
    using VZ = RegVec<2 * EPT_>; using VS = RegVec<2 * EPT_>;
    using VH = BufVec2;  // history vectors and zhat in HBM: 16-byte accesses
};

struct HzPoint {
    double a, v, d;  // alpha, phi(alpha), dphi(alpha)
    int id;          // evaluation sequence number (0 = the point alpha=0)
};

// ------------------------------------------------------------------------------------------------
template <class Model, class Place>
struct Solver {
    static constexpr int T = Place::T, EPT = Place::EPT, U = Place::U, MAXB = Model::MAXB;
    const BatchArgs& a;
    const int tid;
    double* red;
    double* sh_rho;    // [kM]
    double* sh_gam;    // [kM]
    double* sh_alpha;  // [kM]
    int parity;
    typename Place::VX x;
    typename Place::VG g;
    typename Place::VZ z;
    typename Place::VS s;
    double* hist;  // [kM][2][ld] in HBM
    double iv0, sd0;   // MAXB == 1: coefficients in registers
    double* sh_sd;     // MAXB > 1: sampling sd[k] in LDS (the MAP iv[k] is read from the LDS argument block)
    int f_calls;
    int iter_stamp, stamp_p;  // diagnostic build: which evaluation of the first line search
    double last_c, last_gmax;
    // element ownership: thread pairs tfirst + k*pstride (cluster mode: the cluster acts as one csize*T block)
    int tfirst, pstride;
    int crank, csize;          // this workgroup's rank in its cluster, cluster size
    unsigned int cl_epoch;     // cluster reductions done so far in this launch
    bool cl_aborted;           // a bounded wait expired: stop waiting
    unsigned int* cl_counter;  // this cluster's arrival counter
    double* cl_part;           // this cluster's partial slots [2][csize][8]

    __device__ Solver(const BatchArgs& a_, int tid_, double* red_, double* shs) : a(a_), tid(tid_), red(red_), parity(0) {
        sh_rho = shs;
        sh_gam = shs + kM;
        sh_alpha = shs + 2 * kM;
        sh_sd = shs + 3 * kM;
        tfirst = tid_;
        pstride = T;
        crank = 0;
        csize = 1;
        cl_epoch = 0;
        cl_aborted = false;
        cl_counter = nullptr;
        cl_part = nullptr;
    }

    // Cluster all-reduce (Guideline 16 of the CDNA guide: placement-independent release/acquire).
    // Each workgroup has reduced to workgroup-uniform values; lane 0 publishes them with write-through
    // stores, releases at agent scope (which also makes the pass's vector stores visible to the other
    // workgroups of the cluster -- the stencil model reads neighbours across workgroups), arrives on
    // the cluster's monotonic counter and polls it (bounded) for this epoch; one agent-scope acquire,
    // then every thread reads the csize partials with L1-bypassing loads and combines them in rank
    // order.  Partial slots alternate between two buffers by epoch parity (WAR safe).
    template <int KS, int KM>
    __device__ __forceinline__ void cluster_allreduce(double (&sv)[KS > 0 ? KS : 1], double (&mv)[KM > 0 ? KM : 1]) {
        constexpr int K = KS + KM;
        cl_epoch += 1;
        gf64* slots = (gf64*)(cl_part + (size_t)(cl_epoch & 1u) * csize * 8);
        gu32* counter = (gu32*)cl_counter;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
        __syncthreads();
        if (tid == 0) {
            gf64* mine = slots + crank * 8;
#pragma unroll
            for (int k = 0; k < KS; ++k) __hip_atomic_store(mine + k, sv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < KM; ++k) __hip_atomic_store(mine + KS + k, mv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = cl_epoch * (unsigned)csize;
            unsigned spins = 0;
            // bounded (about a second): a cluster whose members are not all resident must not hang the
            // GPU; after one expiry this workgroup never waits again and the host reports the error
            while (!cl_aborted && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 22)) {
                    __hip_atomic_store((gi32*)a.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    cl_aborted = true;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double acc = __hip_atomic_load(slots + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int c = 1; c < csize; ++c) {
                const double v = __hip_atomic_load(slots + c * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc = k < KS ? acc + v : __builtin_fmax(acc, v);
            }
            if (k < KS) sv[k] = uniform(acc);
            else mv[k - KS] = uniform(acc);
        }
    }
    // Workgroup reduction, then (cluster mode) the cluster reduction.
    template <int KS, int KM>
    __device__ __forceinline__ void reduce(double (&sv)[KS > 0 ? KS : 1], double (&mv)[KM > 0 ? KM : 1]) {
        block_allreduce<T, KS, KM>(sv, mv, red, parity, tid);
        if constexpr (Place::kCluster) cluster_allreduce<KS, KM>(sv, mv);
    }
    // Orders this pass's vector stores before the next pass's neighbour reads (stencil model).
    __device__ __forceinline__ void pass_barrier() {
        if constexpr (Place::kCluster) {
            double z1[1] = {0.0}, z2[1] = {0.0};
            cluster_allreduce<1, 0>(z1, z2);
        } else {
            __syncthreads();
        }
    }
    // In-kernel stamps (cdna_hip_programming.md §7): diagnostic build only; values leave through a
    // buffer of their own and no output is computed from them.
    __device__ __forceinline__ void stamp(int p, int k) const {
#ifdef MUSE_STAMPS
        if (tid == 0 && a.stamps) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            a.stamps[(size_t)p * 16 + k] = t;
        }
#else
        (void)p; (void)k;
#endif
    }
    using VH = typename Place::VH;
    __device__ __forceinline__ VH hdx(int slot) const {
        VH v;
        v.bind(hist + (int64_t)(2 * slot) * a.ld, a.ld);
        return v;
    }
    __device__ __forceinline__ VH hdg(int slot) const {
        VH v;
        v.bind(hist + (int64_t)(2 * slot + 1) * a.ld, a.ld);
        return v;
    }
    // Block (theta component) of the element in slot jj / at index i.  Resident policy: a thread's slots are the
    // same elements in every pass of every problem of the launch, so their 3-bit block indices are looked up once
    // per kernel (pk, 10 slots per word) and a pass extracts one with a single v_bfe_u32; streaming: compares.
    unsigned pk[2];
    __device__ __forceinline__ int blk(int jj, int i) const {
        if constexpr (MAXB == 1) return 0;
        else if constexpr (Place::kResident) return (int)((pk[jj / 10] >> (3 * (jj % 10))) & 7u);
        else return block_of<MAXB>(a, i);
    }
    __device__ __forceinline__ double ivk(int jj, int i) const {
        if constexpr (MAXB == 1) return iv0;
        else return a.tmap.iv[blk(jj, i)];
    }
    __device__ __forceinline__ double sdk(int jj, int i) const {
        if constexpr (MAXB == 1) return sd0;
        else return sh_sd[blk(jj, i)];
    }
    __device__ __forceinline__ double sdk_at(int i) const {  // element index only (rolled loops)
        if constexpr (MAXB == 1) return sd0;
        else return sh_sd[block_of<MAXB>(a, i)];
    }
    __device__ __forceinline__ void pack_blocks() {
        pk[0] = pk[1] = 0u;
        if constexpr (MAXB > 1 && Place::kResident) {
            static_assert(2 * EPT <= 20, "two words of ten 3-bit slots");
#pragma unroll
            for (int jj = 0; jj < 2 * EPT; ++jj) {
                const int i = 2 * (tid + (jj >> 1) * T) + (jj & 1);
                pk[jj / 10] |= (unsigned)block_of<MAXB>(a, i) << (3 * (jj % 10));
            }
        }
    }

    // d(-logLike)/dz_i at the point whose components are given by zt(.), for the stencil model
    // (streaming only: neighbours come from HBM/L1).  The element's share of -2 logLike is
    // fma(t, z0, fma(r0, r0, .)) with the returned (t, z0, r0).
    template <class ZT>
    __device__ __forceinline__ double stencil_grad(ZT&& zt, int i, double& t_out, double& z_out, double& r_out) const {
        const int N = (int)a.N;
        auto wrap = [&](int k) { return k < 0 ? k + N : (k >= N ? k - N : k); };
        const int im2 = wrap(i - 2), im1 = wrap(i - 1), ip1 = wrap(i + 1), ip2 = wrap(i + 2);
        const double zm2 = zt(im2), zm1 = zt(im1), z0 = zt(i), zp1 = zt(ip1), zp2 = zt(ip2);
        const double rm = x.get1(im1) - fma(0.25, zm2 + z0, 0.5 * zm1);
        const double r0 = x.get1(i) - fma(0.25, zm1 + zp1, 0.5 * z0);
        const double rp = x.get1(ip1) - fma(0.25, z0 + zp2, 0.5 * zp1);
        const double t = this->ivk(0, i) * z0;
        t_out = t;
        z_out = z0;
        r_out = r0;
        return t - fma(0.25, rm + rp, 0.5 * r0);
    }

    // ---- stencil model, pair-wise --------------------------------------------------------------
    // A lane loads its own element pair of z, s, x with one 16-byte buffer instruction each and gets the
    // neighbouring pairs from the adjacent lanes with DPP wave shifts (lanes of a wave own consecutive
    // pairs); the two edge lanes of a wave fetch their outer neighbour pair from memory with loads that
    // every lane issues but whose offset is out of range (no memory access) for the 62 inner lanes, and
    // the few pairs that touch the periodic wrap or the pad element are patched element-wise afterwards.
    // U pairs per trip: all loads of the trip are issued before anything is stored (see BufChunk).  The
    // arithmetic (operand order included) is that of stencil_grad, and a thread accumulates its elements'
    // shares of the objective in element order, so every path gives the same bits.
    struct Pair {
        double a, b;
    };
    __device__ __forceinline__ static Pair load_pair(const rsrc_t& rs, int i0) {
        Pair p;
        load_f64x2(rs, i0, p.a, p.b);
        return p;
    }
    static constexpr int kDppWaveShr1 = 0x138;  // lane L reads lane L-1
    static constexpr int kDppWaveShl1 = 0x130;  // lane L reads lane L+1
    static constexpr int kOutOfRange = 0x10000000;  // element index beyond any vector (N < 2^28): loads give 0

    // For every pair this thread owns: pre(u, i0) first (the caller's own loads of the trip), then
    // body(u, i0, g0, g1, s0, s1) with g = d(-logLike)/dz at z + c s (at z when !USE_S), zero for the pad
    // element; adds the pairs' shares of -2 logLike to facc; flushes the `written` vectors once per trip.
    template <bool USE_S, class P, class F, class... W>
    __device__ __forceinline__ void stencil_pairs(double c, double& facc, P&& pre, F&& body, W&&... written) {
        const int N = (int)a.N, n = (int)a.ld, lane = tid & 63;
        int t = tfirst;
        asm volatile("" : "+v"(t));
        auto ztf = [&](int k) {
            double v = z.get1(k);
            if constexpr (USE_S) v = fma(c, s.get1(k), v);
            return v;
        };
#pragma unroll 1
        for (int ic = 2 * t; ic < n; ic += 2 * U * pstride) {
            double g0[U], g1[U], tt[U][2], zz[U][2], rr[U][2];
            Pair spv[U];
            bool interior[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i0 = ic + 2 * u * pstride;
                interior[u] = i0 >= 2 && i0 + 3 < N;  // the 6-element window has no wrap and no pad
                const Pair zp = load_pair(z.rsrc, i0), xp = load_pair(x.rsrc, i0);
                // the outer neighbour pair of an edge lane; every other lane aims out of range
                const bool edge = interior[u] && (lane == 0 || lane == 63);
                const int ie = edge ? (lane == 0 ? i0 - 2 : i0 + 2) : kOutOfRange;
                const int ix = edge ? (lane == 0 ? i0 - 1 : i0 + 2) : kOutOfRange;
                Pair ze = load_pair(z.rsrc, ie);
                const double xe = x.get1(ix);
                Pair sp{0.0, 0.0}, ztp = zp;
                if constexpr (USE_S) {
                    sp = load_pair(s.rsrc, i0);
                    const Pair se = load_pair(s.rsrc, ie);
                    ztp = Pair{fma(c, sp.a, zp.a), fma(c, sp.b, zp.b)};
                    ze = Pair{fma(c, se.a, ze.a), fma(c, se.b, ze.b)};
                }
                pre(u, i0);
                spv[u] = sp;
                Pair ztL{dpp_move<kDppWaveShr1>(ztp.a), dpp_move<kDppWaveShr1>(ztp.b)};
                Pair ztR{dpp_move<kDppWaveShl1>(ztp.a), dpp_move<kDppWaveShl1>(ztp.b)};
                double xL = dpp_move<kDppWaveShr1>(xp.b), xR = dpp_move<kDppWaveShl1>(xp.a);
                if (lane == 0) { ztL = ze; xL = xe; }
                if (lane == 63) { ztR = ze; xR = xe; }
                const double rm = xL - fma(0.25, ztL.a + ztp.a, 0.5 * ztL.b);    // r at i0-1
                const double r0 = xp.a - fma(0.25, ztL.b + ztp.b, 0.5 * ztp.a);  // r at i0
                const double r1 = xp.b - fma(0.25, ztp.a + ztR.a, 0.5 * ztp.b);  // r at i0+1
                const double r2 = xR - fma(0.25, ztp.b + ztR.b, 0.5 * ztR.a);    // r at i0+2
                const double t0 = ivk(0, i0) * ztp.a, t1 = ivk(1, i0 + 1) * ztp.b;
                g0[u] = t0 - fma(0.25, rm + r1, 0.5 * r0);
                g1[u] = t1 - fma(0.25, r0 + r2, 0.5 * r1);
                tt[u][0] = t0; zz[u][0] = ztp.a; rr[u][0] = r0;
                tt[u][1] = t1; zz[u][1] = ztp.b; rr[u][1] = r1;
            }
            // wrap-around, pad and out-of-range pairs: element-wise with modular neighbour indices
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i0 = ic + 2 * u * pstride;
                if (!interior[u]) {
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const int i = i0 + v;
                        double gt = 0.0, t_ = 0.0, z_ = 0.0, r_ = 0.0;
                        if (i < N) gt = stencil_grad(ztf, i, t_, z_, r_);
                        (v == 0 ? g0[u] : g1[u]) = gt;
                        tt[u][v] = t_; zz[u][v] = z_; rr[u][v] = r_;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i0 = ic + 2 * u * pstride;
                facc = fma(tt[u][0], zz[u][0], fma(rr[u][0], rr[u][0], facc));
                facc = fma(tt[u][1], zz[u][1], fma(rr[u][1], rr[u][1], facc));
                body(u, i0, g0[u], g1[u], spv[u].a, spv[u].b);
            }
            (written.template flush<U>(ic, pstride), ...);
        }
    }

    // -- objective/gradient at z + c s (or at z when !USE_S).  Returns f = -logLike,
    //    dphi = grad . s and gmax = ||grad||_inf; the gradient itself is stored (into g) only when
    //    STORE_G -- a line-search trial needs just the three scalars.  One pass, one barrier.
    //    INIT_S (resident policy, where s lives in registers): the pass also sets the steepest-descent
    //    direction s = -g and returns dphi = g . s, exactly as the separate pass would.
    template <bool USE_S, bool STORE_G, bool INIT_S = false>
    __device__ __forceinline__ void eval(double c, double& f, double& dphi, double& gmax) {
        double sum[2] = {0.0, 0.0}, mx[1] = {0.0};
        if constexpr (!Model::kStencil) {
            for_elems<T, EPT, U>(a.ld, tfirst, pstride, [&](int jj, int i) {
                double zi = z.get(jj, i);
                double si = 0.0;
                if constexpr (USE_S) {
                    si = s.get(jj, i);
                    zi = fma(c, si, zi);
                }
                const double gi = Model::grad(ivk(jj, i), x.get(jj, i), zi, sum[0]);
                if constexpr (STORE_G) g.set(jj, i, gi);
                if constexpr (USE_S) sum[1] = fma(gi, si, sum[1]);
                if constexpr (INIT_S) {
                    const double sd = -gi;
                    s.set(jj, i, sd);
                    sum[1] = fma(gi, sd, sum[1]);
                }
                mx[0] = absmax(mx[0], gi);
            }, when(STORE_G, g), when(INIT_S, s));
        } else {
            stencil_pairs<USE_S>(c, sum[0], [](int, int) {}, [&](int u, int i0, double g0, double g1, double s0, double s1) {
                if constexpr (STORE_G) {
                    g.set(2 * u, i0, g0);
                    g.set(2 * u + 1, i0 + 1, g1);
                }
                if constexpr (USE_S) sum[1] = fma(g1, s1, fma(g0, s0, sum[1]));
                mx[0] = absmax(absmax(mx[0], g0), g1);
            }, when(STORE_G, g));
        }
        if (iter_stamp == 0) stamp(stamp_p, 14);
        reduce<2, 1>(sum, mx);
        if (iter_stamp == 0) stamp(stamp_p, 15);
        f = 0.5 * (sum[0] + a.f_const);
        dphi = sum[1];
        gmax = nan_if(sum[0] != sum[0] || sum[1] != sum[1], mx[0]);
        f_calls += 1;
    }

    // phi(c), dphi(c) of the line search (the NLSolversBase objective cache is last_c/last_phi).
    __device__ __forceinline__ void phidphi(double c, double& phi, double& dphi) {
        double gm;
        eval<true, false>(c, phi, dphi, gm);
        last_c = c;
        last_gmax = gm;
        last_phi = phi;
    }

    __device__ __forceinline__ static bool finite2(double p, double d) { return isfinite(p) && isfinite(d); }
    __device__ __forceinline__ static double eps_of(double b) {
        const double ab = fabs(b);
        return __longlong_as_double(__double_as_longlong(ab) + 1) - ab;
    }
    __device__ __forceinline__ static double nextup(double v) {  // nextfloat for finite v
        if (v == 0.0) return 4.9406564584124654e-324;
        const long long b = __double_as_longlong(v);
        return __longlong_as_double(v > 0.0 ? b + 1 : b - 1);
    }
    __device__ __forceinline__ static bool wolfe(double c, double phi_c, double dphi_c, double phi_0, double dphi_0,
                                                 double phi_lim) {
        const bool w1 = (kHzDelta * dphi_0 >= (phi_c - phi_0) / c) && (dphi_c >= kHzSigma * dphi_0);
        const bool w2 = ((2.0 * kHzDelta - 1.0) * dphi_0 >= dphi_c) && (dphi_c >= kHzSigma * dphi_0) && (phi_c <= phi_lim);
        return w1 || w2;
    }
    __device__ __forceinline__ static double secant(const HzPoint& p, const HzPoint& q) {
        return (p.a * q.d - q.a * p.d) / (q.d - p.d);
    }

    // HagerZhang line search (LineSearches.jl defaults) from the InitialStatic guess c0, written as a
    // state machine around ONE evaluation site so that the evaluation pass is instantiated once:
    //   S_INIT   first trial (shrunk by psi3 while non-finite)
    //   S_EXPAND bracketing expansion c *= rho           (HZ stages B0-B3)
    //   S_BIS    bisection of [A,B] with dphi(B) < 0      (HZ stage U3, theta = 1/2)
    //   S_SEC1/2 the two secant steps of secant2          (HZ stages S1-S4)
    //   S_MID    bisection step of the main loop when the secant steps shrink too slowly
    //   S_FINAL  re-evaluation at the accepted step when it was not the last trial (update_g!)
    // Points carry an evaluation id so that "which endpoint did update() replace" (HZ U0-U3) is an
    // id comparison, as the index comparisons of the published algorithm.
    // Returns true on success.  On failure (LineSearchException / failed assertion) alpha is the step
    // Optim falls back to.  On success last_phi/last_gmax are the scalars at z + alpha s.
    __device__ bool linesearch(double c0, double phi_0, double dphi_0, double& alpha) {
        enum { S_INIT, S_EXPAND, S_BIS, S_SEC1, S_SEC2, S_MID, S_FINAL };
        enum { CONT_BRACKET, CONT_UPD1, CONT_UPD2, CONT_UPD3 };
        alpha = 0.0;
        if (!finite2(phi_0, dphi_0)) return false;
        if (dphi_0 >= kEps * fabs(phi_0)) return false;  // not a descent direction
        if (dphi_0 >= 0.0 || c0 <= kEps) return true;    // alpha = 0, nothing evaluated
        const double phi_lim = phi_0 + kHzEpsilon * fabs(phi_0);
        const HzPoint P0{0.0, phi_0, dphi_0, 0};
        HzPoint A = P0, B = P0, C = P0, prev = P0, a0 = P0, b0 = P0;
        double alphamax = INFINITY, cold = 0.0, fail_alpha = 0.0, c = c0;
        int state = S_INIT, cont = CONT_BRACKET, iter = 1, iterfinite = 1, nid = 1;
        bool ok = true;
        for (;;) {
            double phi, dphi;
            if (iter_stamp < 3) stamp(stamp_p, 10 + 2 * iter_stamp);
            phidphi(c, phi, dphi);
            if (iter_stamp < 3) stamp(stamp_p, 11 + 2 * iter_stamp);
            iter_stamp += 1;
            const bool fin = finite2(phi, dphi);
            switch (state) {
                case S_INIT:
                    if (!fin) {
                        if (iterfinite < kHzIterFiniteMax) { iterfinite += 1; c *= kHzPsi3; continue; }
                        alpha = 0.0;
                        goto L_DONE;
                    }
                    C = HzPoint{c, phi, dphi, nid++};
                    goto L_BRACKET_TOP;
                case S_EXPAND:
                    if (!fin) {
                        if (c > nextup(cold) && iterfinite < kHzIterFiniteMax) {
                            alphamax = c;
                            iterfinite += 1;
                            c = (cold + c) / 2.0;
                            continue;
                        }
                        alpha = cold;
                        goto L_DONE;
                    }
                    C = HzPoint{c, phi, dphi, nid++};
                    iter += 1;
                    goto L_BRACKET_TOP;
                case S_BIS: {
                    if (!fin) goto L_FAIL;
                    const HzPoint D{c, phi, dphi, nid++};
                    if (D.d >= 0.0) { B = D; goto L_BISECT_RETURN; }
                    if (D.v <= phi_lim) A = D;
                    else B = D;
                    goto L_BISECT_TOP;
                }
                case S_SEC1:
                    if (!fin) goto L_FAIL;
                    C = HzPoint{c, phi, dphi, nid++};
                    if (wolfe(c, phi, dphi, phi_0, dphi_0, phi_lim)) { alpha = c; goto L_DONE; }
                    cont = CONT_UPD1;
                    goto L_UPDATE;
                case S_SEC2:
                    if (!fin) goto L_FAIL;
                    C = HzPoint{c, phi, dphi, nid++};
                    if (wolfe(c, phi, dphi, phi_0, dphi_0, phi_lim)) { alpha = c; goto L_DONE; }
                    cont = CONT_UPD2;
                    goto L_UPDATE;
                case S_MID:
                    if (!fin) goto L_FAIL;
                    C = HzPoint{c, phi, dphi, nid++};
                    cont = CONT_UPD3;
                    goto L_UPDATE;
                default:  // S_FINAL
                    return ok;
            }
        L_BRACKET_TOP:
            if (!(iter < kHzLinesearchMax)) { alpha = 0.0; goto L_FAIL_KEEP; }  // never bracketed
            if (C.d >= 0.0) {
                A = prev;  // the latest point with phi <= phi_lim and dphi < 0 (HZ look-back)
                B = C;
                iter += 1;
                goto L_MAIN_TOP;
            } else if (C.v > phi_lim) {
                A = P0;
                B = C;
                cont = CONT_BRACKET;
                fail_alpha = 0.0;
                goto L_BISECT_ENTER;
            } else {
                cold = C.a;
                if (nextup(cold) >= alphamax) { alpha = cold; goto L_DONE; }
                prev = C;
                c = cold * kHzRho;
                if (c > alphamax) c = alphamax;
                iterfinite = 1;
                state = S_EXPAND;
                continue;
            }
        L_UPDATE:  // HZ stages U0-U3 on (A, B) with the new point C
            if (!(A.d < 0.0 && A.v <= phi_lim && B.d >= 0.0 && B.a > A.a)) goto L_FAIL;
            if (C.a < A.a || C.a > B.a) goto L_UPDATE_RETURN;
            if (C.d >= 0.0) { B = C; goto L_UPDATE_RETURN; }
            if (C.v <= phi_lim) { A = C; goto L_UPDATE_RETURN; }
            B = C;
        L_BISECT_ENTER:
            if (!(A.d < 0.0 && A.v <= phi_lim && B.d < 0.0 && B.v > phi_lim && B.a > A.a)) goto L_FAIL;
        L_BISECT_TOP:
            if (B.a - A.a > eps_of(B.a)) {
                c = (A.a + B.a) / 2.0;
                state = S_BIS;
                continue;
            }
        L_BISECT_RETURN:
            if (cont == CONT_BRACKET) { iter += 1; goto L_MAIN_TOP; }
        L_UPDATE_RETURN:
            if (cont == CONT_UPD1) {
                const bool updB = (B.id == C.id), updA = (A.id == C.id);
                double c2 = C.a;
                if (updB) c2 = secant(b0, B);
                else if (updA) c2 = secant(a0, A);
                if ((updA || updB) && A.a <= c2 && c2 <= B.a) {
                    c = c2;
                    state = S_SEC2;
                    continue;
                }
            }
            if (cont == CONT_UPD3) { iter += 1; goto L_MAIN_TOP; }
            // after secant2 (CONT_UPD1 without a second step, or CONT_UPD2)
            if (!(B.a > A.a)) { alpha = A.a; goto L_FAIL_KEEP; }
            if (B.a - A.a < kHzGamma * (b0.a - a0.a)) {
                if (nextup(a0.v) >= b0.v && nextup(A.v) >= B.v) { alpha = A.a; goto L_DONE; }  // flat
                iter += 1;
                goto L_MAIN_TOP;
            }
            fail_alpha = A.a;
            c = (A.a + B.a) / 2.0;
            state = S_MID;
            continue;
        L_MAIN_TOP:
            if (!(iter < kHzLinesearchMax)) { alpha = A.a; goto L_FAIL_KEEP; }
            a0 = A;
            b0 = B;
            if (!(b0.a > a0.a)) { alpha = a0.a; goto L_FAIL_KEEP; }
            if (b0.a - a0.a <= eps_of(b0.a)) { alpha = a0.a; goto L_DONE; }
            fail_alpha = a0.a;
            if (!(a0.d < 0.0 && b0.d >= 0.0)) goto L_FAIL;
            c = secant(a0, b0);
            if (!isfinite(c)) goto L_FAIL;
            state = S_SEC1;
            continue;
        L_FAIL:
            alpha = fail_alpha;
        L_FAIL_KEEP:
            ok = false;
        L_DONE:
            // update_g!: NLSolversBase re-evaluates unless z + alpha s is the point evaluated last
            if (!ok) return false;
            if (alpha == last_c) return true;
            c = alpha;
            state = S_FINAL;
        }
    }

    // -- one element: sample/load x, MAP by L-BFGS, score.
    // state of the element being processed, shared by the phases begin -> solve -> finish
    ProblemDesc d;
    double f, gmax;
    int iterations, hist_words, status;
    double* extra;  // one more scratch vector (streaming): the simulation's true z for the implicit-diff H

    __device__ void run(int p, double* wg_scratch, double* lds_x, double* lds_g) {
        begin<false>(p, wg_scratch, lds_x, lds_g);
        if (d.normals_only) return;  // the element only filled its slot of the normals cache
        solve(p);
        finish(p);
    }

    // -- phase 1: bind storage, produce x and the starting point
    template <bool KEEP_ZTRUE>
    __device__ void begin(int p, double* wg_scratch, double* lds_x, double* lds_g) {
        d = describe(a, p);
        const int64_t N = a.N, ld = a.ld;
        stamp(p, 0);
        iv0 = a.tmap.iv[0];
        sd0 = d.tsample >= 0 ? a.tsample[d.tsample].sd[0] : a.tmap.sd[0];
        if constexpr (MAXB > 1) {
            // FD batches sample at a theta that differs from the MAP theta
            if (tid < MAXB) sh_sd[tid] = d.tsample >= 0 ? a.tsample[d.tsample].sd[tid] : a.tmap.sd[tid];
            __syncthreads();
        }
        // bind storage
        const double* zmem = nullptr;
        if constexpr (Place::kResident) {
            hist = wg_scratch;
            if constexpr (Place::kXgLds) {
                x.bind(lds_x, ld);
                g.bind(lds_g, ld);
            }
        } else {
            x.bind(wg_scratch, ld);
            g.bind(wg_scratch + ld, ld);
            s.bind(wg_scratch + 2 * ld, ld);
            zmem = d.zslot >= 0 ? a.zhat + d.zslot * ld : wg_scratch + 3 * ld;
            z.bind(zmem, ld);
            hist = wg_scratch + 4 * ld;
            extra = hist + (int64_t)2 * kM * ld;
        }
        const double* z0ptr = a.zhat + d.z0slot * ld;
        VH z0src;
        z0src.bind(z0ptr, ld);
        const bool z_in_place = (!Place::kResident) && (d.z0_mode == Z0_WARM || d.z0_mode == Z0_COPY) && (z0ptr == zmem);

        // ---- x and the starting point ---------------------------------------------------------
        if (d.x_mode == X_SAMPLE) {
            const uint64_t sim = (uint64_t)d.sim;
            if constexpr (Place::kResident && Place::kXgLds) {
                // Sampler as a ROLLED loop over this thread's pairs (one or two Philox/Box-Muller
                // chains in flight, not 2*EPT): x goes straight to LDS, the true z is staged in the
                // (still unused) g area and picked up into registers below.
                const int nmode = d.nslot >= 0 ? a.ncache_mode : 0;  // workgroup-uniform
                rsrc_t n1r = make_rsrc(a.ncache, 0), n2r = n1r;
                if (nmode != 0) {
                    n1r = make_rsrc(a.ncache + (int64_t)(2 * d.nslot) * ld, ld * 8);
                    n2r = make_rsrc(a.ncache + (int64_t)(2 * d.nslot + 1) * ld, ld * 8);
                }
                if (nmode == 2) {
                    // the stream was drawn earlier in this host call: its normals come from HBM (all of the
                    // thread's loads in flight at once; out-of-range pairs read zeros), not from the generator
                    double c1[EPT][2], c2[EPT][2];
#pragma unroll
                    for (int j = 0; j < EPT; ++j) {
                        const int i0 = 2 * (tid + j * T);
                        load_f64x2(n1r, i0, c1[j][0], c1[j][1]);
                        load_f64x2(n2r, i0, c2[j][0], c2[j][1]);
                    }
#pragma unroll
                    for (int j = 0; j < EPT; ++j) {
                        const int i0 = 2 * (tid + j * T);
                        double zt0, xt0, zt1, xt1;
                        Model::sample(sdk(2 * j, i0), c1[j][0], c2[j][0], zt0, xt0);
                        Model::sample(sdk(2 * j + 1, i0 + 1), c1[j][1], c2[j][1], zt1, xt1);
                        const bool valid1 = i0 + 1 < (int)N;
                        x.set(2 * j, i0, xt0);
                        x.set(2 * j + 1, i0 + 1, valid1 ? xt1 : 0.0);
                        g.set(2 * j, i0, zt0);
                        g.set(2 * j + 1, i0 + 1, valid1 ? zt1 : 0.0);
                    }
                } else {
#pragma unroll 1
                    for (int i0 = 2 * tid; i0 < (int)N; i0 += 2 * T) {
                        // both elements of the pair unconditionally (one basic block: their Philox/Box-Muller
                        // chains interleave); for odd N the last pair's second element is the pad slot, kept at 0
                        const NormalPair np0 = normal_pair(a.seed, sim, (uint64_t)i0);
                        const NormalPair np1 = normal_pair(a.seed, sim, (uint64_t)(i0 + 1));
                        if (nmode == 1) {
                            store_f64x2(n1r, i0, np0.n1, np1.n1);
                            store_f64x2(n2r, i0, np0.n2, np1.n2);
                        }
                        double zt0, xt0, zt1, xt1;
                        Model::sample(sdk_at(i0), np0.n1, np0.n2, zt0, xt0);
                        Model::sample(sdk_at(i0 + 1), np1.n1, np1.n2, zt1, xt1);
                        const bool valid1 = i0 + 1 < (int)N;
                        x.p[i0] = xt0;
                        g.p[i0] = zt0;
                        x.p[i0 + 1] = valid1 ? xt1 : 0.0;
                        g.p[i0 + 1] = valid1 ? zt1 : 0.0;
                    }
                }
                z.clear();
                s.clear();
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    if (d.z0_mode == Z0_ZERO) z.set(jj, i, 0.0);
                    else if (d.z0_mode == Z0_TRUE) z.set(jj, i, g.get(jj, i));
                    else z.set(jj, i, z0src.get(jj, i));
                });
            } else {
                x.clear(); g.clear(); z.clear(); s.clear();
                VH ztrue;
                if constexpr (KEEP_ZTRUE) ztrue.bind(extra, ld);
                // streaming: ONE pair per trip (two Philox/Box-Muller chains in flight, as in the rolled
                // resident sampler) -- the pass is generator-bound and loads nothing; a warm start is
                // copied by a pass of its own below
                constexpr int US = Place::kResident ? U : 1;
                const bool z_from_sample = d.z0_mode == Z0_ZERO || d.z0_mode == Z0_TRUE;
                for_elems<T, EPT, US>(ld, tfirst, pstride, [&](int jj, int i) {
                    const bool valid = i < N;  // phantom slots run the generator but keep zeros
                    const NormalPair np = normal_pair(a.seed, sim, (uint64_t)i);
                    double zt, xt;
                    if constexpr (Model::kStencil) {
                        zt = sdk(jj, i) * np.n1;
                        xt = np.n2;                           // noise now, + A z after the barrier
                        s.set(jj, i, valid ? zt : 0.0);       // true z staged in the direction buffer
                    } else {
                        Model::sample(sdk(jj, i), np.n1, np.n2, zt, xt);
                    }
                    zt = valid ? zt : 0.0;
                    xt = valid ? xt : 0.0;
                    if constexpr (KEEP_ZTRUE) ztrue.set(jj, i, zt);
                    x.set(jj, i, xt);
                    if (d.z0_mode == Z0_ZERO) z.set(jj, i, 0.0);
                    else if (d.z0_mode == Z0_TRUE) z.set(jj, i, zt);
                    else if constexpr (Place::kResident) z.set(jj, i, z0src.get(jj, i));
                }, when(Model::kStencil, s), when(KEEP_ZTRUE, ztrue), x, when(z_from_sample, z));
                if constexpr (!Place::kResident) {
                    if (!z_from_sample && !z_in_place)
                        for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) { z.set(jj, i, z0src.get(jj, i)); }, z);
                }
            }
            if constexpr (Model::kStencil) {
                pass_barrier();
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const bool valid = i < N;
                    const int ic = valid ? i : 0;
                    const int im = ic == 0 ? (int)N - 1 : ic - 1, ip = ic == (int)N - 1 ? 0 : ic + 1;
                    const double az = fma(0.25, s.get1(im) + s.get1(ip), 0.5 * s.get1(ic));
                    const double xv = az + x.get(jj, i);
                    x.set(jj, i, valid ? xv : 0.0);
                }, x);
            }
        } else {
            VH xs;
            xs.bind(d.x_mode == X_DATA ? a.x_data : a.x_given, ld);
            x.clear(); g.clear(); z.clear(); s.clear();
            const bool z_zero = d.z0_mode == Z0_ZERO || d.z0_mode == Z0_TRUE;
            for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                x.set(jj, i, xs.get(jj, i));
                if (z_zero) z.set(jj, i, 0.0);
                else if (!z_in_place) z.set(jj, i, z0src.get(jj, i));
            }, x, when(z_zero || !z_in_place, z));
        }
        stamp(p, 9);
        if constexpr (Model::kStencil) pass_barrier();  // x (and the start z) complete before neighbours read them
        else __syncthreads();

        stamp(p, 1);
    }

    // -- phase 2: zhat_at_theta -- Optim LBFGS + HagerZhang on -logLike from the z prepared by begin()
    __device__ void solve(int p) {
        const int64_t ld = a.ld;
        const int N = (int)a.N;
        // ---- initial_state: value_gradient!!(d, z0); initial convergence -----------------------
        f_calls = 0;
        iter_stamp = 99;
        stamp_p = p;
        last_c = NAN;
        // Resident policy, elementwise models: the initial evaluation leaves s = -g and g . s behind and does NOT
        // store g (in LDS a store between the loads serialises the pass); the first update pass that needs the old
        // gradient recomputes it from (x, z) -- bit-identical -- and stores the new one.  A solve that ends with
        // its first line search (every isotropic problem) never writes or reads g at all.
        constexpr bool kFuseInit = Place::kResident && !Model::kStencil;
        double dphi_init;
        eval<false, !kFuseInit, kFuseInit>(0.0, f, dphi_init, gmax);
        bool g_stored = !kFuseInit;
        score_ready = false;
        stamp(p, 2);
        iterations = 0;
        hist_words = 0;
        int pseudo = 0, counter_f_tol = 0;
        status = MUSE_STATUS_MAXITER;
        bool done = false;
        if (!isfinite(f) || !isfinite(gmax)) { status = MUSE_STATUS_NONFINITE; done = true; }
        else if (gmax <= a.atol) { status = MUSE_STATUS_G_CONVERGED; done = true; }
        if (a.debug & 1) done = true;

        double dot0 = 0.0;      // dot(dx_newest, g) prepared by the update pass
        bool have_pair = false;  // the update pass of the previous iteration stored a usable pair
        while (!done && iterations < kMaxIter) {
            iterations += 1;
            pseudo += 1;
            // ---- twoloop!: s = -H g ------------------------------------------------------------
            const int upper = pseudo - 1, lower = (pseudo - kM) > 1 ? (pseudo - kM) : 1;
            const int h = upper >= lower ? upper - lower + 1 : 0;
            double dphi_0;
            if (kFuseInit && iterations == 1) {
                dphi_0 = dphi_init;  // s = -g and g . s came with the initial evaluation
            } else if (h == 0 || !have_pair) {
                double sum[1] = {0.0}, mx[1] = {0.0};
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double gi = g.get(jj, i), si = -gi;
                    s.set(jj, i, si);
                    sum[0] = fma(gi, si, sum[0]);
                }, s);
                reduce<1, 0>(sum, mx);
                dphi_0 = sum[0];
            } else {
                hist_words += h;
                double dot = dot0;
                // backward pass (q lives in s; the update pass left q = g there)
                for (int index = upper; index >= lower; --index) {
                    const int slot = (index - 1) % kM;
                    const double al = sh_rho[slot] * dot;
                    if ((tid & 63) == 0) sh_alpha[slot] = al;  // lane 0 of EVERY wave: a wave reads back its own write (no barrier needed)
                    const VH dgp = hdg(slot);
                    double sum[1] = {0.0}, mx[1] = {0.0};
                    if (index > lower) {
                        const VH dxn = hdx((index - 2) % kM);
                        for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                            const double qi = fma(-al, dgp.get(jj, i), s.get(jj, i));
                            s.set(jj, i, qi);
                            sum[0] = fma(dxn.get(jj, i), qi, sum[0]);
                        }, s);
                    } else {  // last backward step: apply gamma = (dx.dg)/(dg.dg) of the newest pair
                        const double gam = sh_gam[(upper - 1) % kM];
                        for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                            const double dgi = dgp.get(jj, i);
                            const double si = gam * fma(-al, dgi, s.get(jj, i));
                            s.set(jj, i, si);
                            sum[0] = fma(dgi, si, sum[0]);
                        }, s);
                    }
                    reduce<1, 0>(sum, mx);
                    dot = sum[0];
                }
                // forward pass
                for (int index = lower; index <= upper; ++index) {
                    const int slot = (index - 1) % kM;
                    const double beta = sh_rho[slot] * dot;
                    const double coef = sh_alpha[slot] - beta;
                    const VH dxp = hdx(slot);
                    double sum[1] = {0.0}, mx[1] = {0.0};
                    if (index < upper) {
                        const VH dgn = hdg(index % kM);
                        for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                            const double si = fma(dxp.get(jj, i), coef, s.get(jj, i));
                            s.set(jj, i, si);
                            sum[0] = fma(dgn.get(jj, i), si, sum[0]);
                        }, s);
                    } else {
                        for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                            const double si = -fma(dxp.get(jj, i), coef, s.get(jj, i));
                            s.set(jj, i, si);
                            sum[0] = fma(g.get(jj, i), si, sum[0]);
                        }, s);
                    }
                    reduce<1, 0>(sum, mx);
                    dot = sum[0];
                }
                dphi_0 = dot;
            }
            // ---- perform_linesearch!: reset a non-descent direction ------------------------------
            if (dphi_0 >= 0.0) {
                pseudo = 1;
                if constexpr (kFuseInit) {
                    if (!g_stored) {  // (cannot happen after a finite, unconverged initial evaluation; kept for completeness)
                        for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                            double unused = 0.0;
                            g.set(jj, i, Model::grad(ivk(jj, i), x.get(jj, i), z.get(jj, i), unused));
                        }, g);
                        g_stored = true;
                    }
                }
                double sum[1] = {0.0}, mx[1] = {0.0};
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double gi = g.get(jj, i), si = -gi;
                    s.set(jj, i, si);
                    sum[0] = fma(gi, si, sum[0]);
                }, s);
                reduce<1, 0>(sum, mx);
                dphi_0 = sum[0];
            }
            if (iterations == 1) stamp(p, 3);
            const double phi_0 = f, f_prev = f;
            last_c = NAN;  // no trial evaluated yet in this line search
            last_phi = f;
            last_gmax = gmax;
            double alpha;
            iter_stamp = iterations == 1 ? 0 : 99;
            stamp_p = p;
            const bool ls_ok = linesearch(1.0, phi_0, dphi_0, alpha);
            if (iterations == 1) stamp(p, 4);
            // ---- update_g! / assess_convergence: the scalars at z + alpha s are those of the last
            //      evaluation (or of the current point when the step is a no-op) ---------------------
            const double f_new = last_phi, gmax_new = last_gmax;
            const bool g_conv = gmax_new <= a.atol;
            const bool f_conv = fabs(f_new - f_prev) <= 0.0;
            const int cft = f_conv ? counter_f_tol + 1 : 0;
            const bool stop_hint = !ls_ok || g_conv || cft > 1 || !isfinite(gmax_new);
            // ---- fused update pass: z += alpha s; gradient at the new point recomputed (bit-equal
            //      to the trial evaluation); (dx, dg) stored; g <- new gradient; q <- g -----------------
            const int slot_new = (pseudo - 1) % kM;
            VH dxs = hdx(slot_new);
            VH dgs = hdg(slot_new);
            const bool keep = !stop_hint;
            double sum[3] = {0.0, 0.0, 0.0}, mx[1] = {0.0};
            if (!keep) {
                // ---- the solve ends with this step (whatever x_converged says): z += alpha s, and in the same
                //      pass what finish() would compute from the final z -- the score terms and the zhat store
                double acc[MAXB];
#pragma unroll
                for (int b = 0; b < MAXB; ++b) acc[b] = 0.0;
                VH zout;
                const bool store = Place::kResident && d.zslot >= 0;  // streaming: z already lives in its zhat slot
                if (store) zout.bind(a.zhat + d.zslot * ld, ld);
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double zo = z.get(jj, i), si = s.get(jj, i);
                    const double zn = fma(alpha, si, zo);
                    z.set(jj, i, zn);
                    mx[0] = absmax(mx[0], zn - zo);
                    if constexpr (Place::kResident) {
                        if (store) zout.set(jj, i, zn);
                    }
                    const double t = Model::score_term(x.get(jj, i), zn);
                    if constexpr (MAXB == 1) {
                        acc[0] += t;
                    } else {
                        const int k = blk(jj, i);
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) acc[b] += (k == b) ? t : 0.0;
                    }
                }, z);
                if constexpr (MAXB + 1 <= 8) {
                    reduce<MAXB, 1>(acc, mx);
                } else {
                    double none[1] = {0.0};
                    reduce<0, 1>(none, mx);
                    reduce<MAXB, 0>(acc, none);
                }
                bool any_nan = false;  // a NaN step shows up in the score sums (NaN channel of the maximum)
#pragma unroll
                for (int b = 0; b < MAXB; ++b) {
                    score_acc[b] = acc[b];
                    any_nan = any_nan || acc[b] != acc[b];
                }
                mx[0] = nan_if(any_nan, mx[0]);
                score_ready = true;
            } else if constexpr (!Model::kStencil) {
                auto body = [&](auto have_g, int jj, int i) {
                    const double zo = z.get(jj, i), si = s.get(jj, i);
                    const double dxi = alpha * si;
                    const double zn = fma(alpha, si, zo);  // the same point the accepted trial evaluated
                    z.set(jj, i, zn);
                    mx[0] = absmax(mx[0], zn - zo);
                    double unused = 0.0;
                    const double xi = x.get(jj, i), ivi = ivk(jj, i);
                    const double gn = Model::grad(ivi, xi, zn, unused);
                    double go;
                    if constexpr (decltype(have_g)::value) go = g.get(jj, i);
                    else go = Model::grad(ivi, xi, zo, unused);  // what the initial evaluation computed
                    const double dgi = gn - go;
                    sum[0] = fma(dxi, dgi, sum[0]);
                    sum[1] = fma(dgi, dgi, sum[1]);
                    sum[2] = fma(dxi, gn, sum[2]);
                    dxs.set(jj, i, dxi);
                    dgs.set(jj, i, dgi);
                    g.set(jj, i, gn);
                    s.set(jj, i, gn);
                };
                if (g_stored) {
                    for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) { body(std::true_type{}, jj, i); },
                                         z, dxs, dgs, g, s);
                } else {
                    if constexpr (kFuseInit)
                        for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) { body(std::false_type{}, jj, i); },
                                             z, dxs, dgs, g, s);
                    g_stored = true;
                }
                reduce<3, 1>(sum, mx);
                mx[0] = nan_if(sum[0] != sum[0], mx[0]);
            } else {
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double zo = z.get(jj, i), si = s.get(jj, i);
                    const double dxi = alpha * si;
                    const double zn = fma(alpha, si, zo);
                    z.set(jj, i, zn);
                    mx[0] = absmax(mx[0], zn - zo);
                    dxs.set(jj, i, dxi);
                }, z, dxs);
                pass_barrier();  // neighbours' z must be complete before the stencil reads them
                double unused = 0.0;
                double dxv[U][2], gov[U][2];
                stencil_pairs<false>(0.0, unused,
                    [&](int u, int i0) {  // this trip's own loads, issued with the stencil's
                        dxv[u][0] = dxs.get(2 * u, i0);
                        dxv[u][1] = dxs.get(2 * u + 1, i0 + 1);
                        gov[u][0] = g.get(2 * u, i0);
                        gov[u][1] = g.get(2 * u + 1, i0 + 1);
                    },
                    [&](int u, int i0, double gn0, double gn1, double, double) {
                        const double gnv[2] = {gn0, gn1};
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const int i = i0 + v;
                            const double gn = gnv[v];
                            const double dxi = dxv[u][v];
                            const double dgi = gn - gov[u][v];
                            sum[0] = fma(dxi, dgi, sum[0]);
                            sum[1] = fma(dgi, dgi, sum[1]);
                            sum[2] = fma(dxi, gn, sum[2]);
                            dgs.set(2 * u + v, i, dgi);
                            g.set(2 * u + v, i, gn);
                            s.set(2 * u + v, i, gn);
                        }
                    }, dgs, g, s);
                reduce<3, 1>(sum, mx);
                mx[0] = nan_if(sum[0] != sum[0], mx[0]);
            }
            if (iterations == 1) stamp(p, 5);
            if (!ls_ok) {  // Optim keeps value(d)/gradient(d) of the last evaluated point
                status = MUSE_STATUS_LINESEARCH_FAILED;
                f = last_phi;
                gmax = last_gmax;
                break;
            }
            f = f_new;
            gmax = gmax_new;
            counter_f_tol = cft;
            const bool x_conv = mx[0] <= 0.0;
            done = stop_hint || x_conv;
            if (g_conv) status = MUSE_STATUS_G_CONVERGED;
            else if (x_conv) status = MUSE_STATUS_X_CONVERGED;
            else if (cft > 1) status = MUSE_STATUS_F_CONVERGED;
            if (!isfinite(gmax)) status = MUSE_STATUS_NONFINITE;
            // ---- update_h!: rho = 1/(dx.dg); an infinite rho drops the history ----------------------
            have_pair = false;
            if (keep) {
                const double rho_it = 1.0 / sum[0];
                if (isinf(rho_it)) {
                    pseudo = 0;
                } else {
                    if ((tid & 63) == 0) {  // every wave writes the identical value and later reads its own write
                        sh_rho[slot_new] = rho_it;
                        sh_gam[slot_new] = sum[0] / sum[1];
                    }
                    have_pair = true;
                }
                dot0 = sum[2];
            }
        }

        stamp(p, 6);
    }

    // ------------------------------------------------------------------------------------------
    // get_H! implicit-differentiation branch for one simulation (src/muse.jl:335-405):
    //   H = H1 - dFdtheta^T A^{-1} dFdtheta1,  A = Hessian_z logLike at (x, zhat, theta0),
    // A^{-1} by conjugate gradients (IterativeSolvers.cg: x0 = 0, reltol sqrt(eps), abstol 0, maxiter).
    // The reference gets the derivative operands by nested AD; for the compiled-in models they are
    // closed forms (see oracle/muse_oracle.c, mo_implicit_H, for the list).  Streaming policy only:
    // the CG vectors reuse the solver's g, s and history buffers; z_true sits in the extra vector.
    // Writes H[p] (row-major ntheta x ntheta) and the CG iteration count of column j to info[p*ntheta+j].
    __device__ void run_implicit(int p, double* wg_scratch, double* lds_x, double* lds_g) {
        begin<true>(p, wg_scratch, lds_x, lds_g);
        solve(p);
        const int64_t ld = a.ld;
        const int N = (int)a.N, nth = a.ntheta;
        VH ztrue, v, r, pp, Ap, t1, t2;
        ztrue.bind(extra, ld);
        v.bind(wg_scratch + ld, ld);       // g buffer
        r.bind(wg_scratch + 2 * ld, ld);   // s buffer
        pp.bind(hist, ld);
        Ap.bind(hist + ld, ld);
        t1.bind(hist + 2 * ld, ld);
        t2.bind(hist + 3 * ld, ld);
        auto Aat = [&](const VH& w, int i) {  // (A w)_i, periodic (1/4, 1/2, 1/4); the pad element maps to 0
            const bool valid = i < N;
            const int ic = valid ? i : 0;
            const int im = ic == 0 ? N - 1 : ic - 1, ip = ic == N - 1 ? 0 : ic + 1;
            const double a0 = fma(0.25, w.get1(im) + w.get1(ip), 0.5 * w.get1(ic));
            return valid ? a0 : 0.0;
        };
        for (int j = 0; j < nth; ++j) {
            // ---- right-hand side b = dFdtheta1[:, j]; v = 0, r = p = b --------------------------------
            double sum[1] = {0.0}, mx[1] = {0.0};
            if constexpr (Model::kStencil) {
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double zt = ztrue.get(jj, i);  // unconditional: the pair load is issued at the even element
                    t1.set(jj, i, blk(jj, i) == j ? 0.5 * zt : 0.0);
                }, t1);
                pass_barrier();
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) { t2.set(jj, i, Aat(t1, i)); }, t2);
                pass_barrier();
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double bi = Aat(t2, i);
                    v.set(jj, i, 0.0);
                    r.set(jj, i, bi);
                    pp.set(jj, i, bi);
                    sum[0] = fma(bi, bi, sum[0]);
                }, v, r, pp);
            } else {
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double zt = ztrue.get(jj, i);  // unconditional: the pair load is issued at the even element
                    double bi;
                    if constexpr (Model::kId == MUSE_MODEL_NOISE) bi = iv0 * (0.5 * (x.get(jj, i) - zt));
                    else bi = blk(jj, i) == j ? 0.5 * zt : 0.0;
                    v.set(jj, i, 0.0);
                    r.set(jj, i, bi);
                    pp.set(jj, i, bi);
                    sum[0] = fma(bi, bi, sum[0]);
                }, v, r, pp);
            }
            reduce<1, 0>(sum, mx);
            double rr = sum[0];
            const double tol = __builtin_sqrt(kEps) * __builtin_sqrt(rr);
            int it = 0;
            while (it < a.cg_maxiter && !(__builtin_sqrt(rr) <= tol)) {
                // ---- Ap = A_hess p, p.Ap -----------------------------------------------------------
                double s1[1] = {0.0};
                if constexpr (Model::kStencil) {
                    for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) { t1.set(jj, i, Aat(pp, i)); }, t1);
                    pass_barrier();
                    for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                        const double pi = pp.get(jj, i);
                        const double api = -(Aat(t1, i) + ivk(jj, i) * pi);
                        Ap.set(jj, i, api);
                        s1[0] = fma(pi, api, s1[0]);
                    }, Ap);
                } else {
                    for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                        const double pi = pp.get(jj, i);
                        double api;
                        if constexpr (Model::kId == MUSE_MODEL_NOISE) api = -((iv0 + 1.0) * pi);
                        else api = -(pi + ivk(jj, i) * pi);
                        Ap.set(jj, i, api);
                        s1[0] = fma(pi, api, s1[0]);
                    }, Ap);
                }
                reduce<1, 0>(s1, mx);
                const double alpha = rr / s1[0];
                // ---- v += alpha p ; r -= alpha Ap ; r.r ---------------------------------------------
                double s2[1] = {0.0};
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    v.set(jj, i, fma(alpha, pp.get(jj, i), v.get(jj, i)));
                    const double ri = fma(-alpha, Ap.get(jj, i), r.get(jj, i));
                    r.set(jj, i, ri);
                    s2[0] = fma(ri, ri, s2[0]);
                }, v, r);
                reduce<1, 0>(s2, mx);
                const double beta = s2[0] / rr;
                rr = s2[0];
                // ---- p = r + beta p (its stores are ordered before the next stencil read by pass_barrier)
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    pp.set(jj, i, fma(beta, pp.get(jj, i), r.get(jj, i)));
                }, pp);
                if constexpr (Model::kStencil) pass_barrier();
                it += 1;
            }
            // ---- H[:, j] = H1[:, j] - dFdtheta^T v ----------------------------------------------------
            constexpr int KA = Model::kId == MUSE_MODEL_NOISE ? 2 : MAXB;  // noise: [dFdtheta^T v, H1 sum]
            double acc[KA];
#pragma unroll
            for (int b = 0; b < KA; ++b) acc[b] = 0.0;
            for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                const double zi = z.get(jj, i), vi = v.get(jj, i);
                if constexpr (Model::kId == MUSE_MODEL_NOISE) {
                    const double xi = x.get(jj, i);
                    const double dd = xi - zi;
                    acc[0] = fma(-iv0 * dd, vi, acc[0]);
                    acc[1] = fma(dd, 0.5 * (xi - ztrue.get(jj, i)), acc[1]);
                } else {
                    const double t = ivk(jj, i) * zi;
                    if constexpr (MAXB == 1) {
                        acc[0] = fma(t, vi, acc[0]);
                    } else {
                        const int k = blk(jj, i);
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) acc[b] = (k == b) ? fma(t, vi, acc[b]) : acc[b];
                    }
                }
            });
            reduce<KA, 0>(acc, mx);
            if (tid == 0 && crank == 0) {
#pragma unroll
                for (int b = 0; b < MAXB; ++b) {
                    if (b < nth) {
                        double h1 = 0.0;
                        if constexpr (Model::kId == MUSE_MODEL_NOISE) h1 = iv0 * acc[1];
                        a.scores[((int64_t)p * nth + b) * nth + j] = h1 - acc[b];
                    }
                }
                muse_info inf;
                inf.iterations = it;
                inf.f_calls = f_calls;
                inf.status = status;
                inf.hist_words = hist_words;
                inf.f_min = f;
                inf.gnorm = gmax;
                a.info[(int64_t)p * nth + j] = inf;
            }
        }
    }

    // -- phase 3: zhat out, score grad_theta logLike(x, zhat, theta), solver info
    __device__ void finish(int p) {
        const int64_t ld = a.ld;
        {
            double acc[MAXB], mx[1] = {0.0};
            if (score_ready) {  // the solve's last pass already did both (see solve())
#pragma unroll
                for (int b = 0; b < MAXB; ++b) acc[b] = score_acc[b];
            } else {
                VH zout;
                const bool store = Place::kResident && d.zslot >= 0;
                if (store) zout.bind(a.zhat + d.zslot * ld, ld);
#pragma unroll
                for (int b = 0; b < MAXB; ++b) acc[b] = 0.0;
                for_elems<T, EPT, U>(ld, tfirst, pstride, [&](int jj, int i) {
                    const double zi = z.get(jj, i);
                    if constexpr (Place::kResident) {
                        if (store) zout.set(jj, i, zi);
                    }
                    const double t = Model::score_term(x.get(jj, i), zi);
                    if constexpr (MAXB == 1) {
                        acc[0] += t;
                    } else {
                        const int k = blk(jj, i);
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) acc[b] += (k == b) ? t : 0.0;
                    }
                });
                reduce<MAXB, 0>(acc, mx);
            }
            if (tid < MAXB && tid < a.ntheta && crank == 0) {  // lane b finishes and writes score component b
                double mine = acc[0];
#pragma unroll
                for (int b = 1; b < MAXB; ++b) mine = (tid == b) ? acc[b] : mine;
                const double cnt = (double)(a.bnd32[tid < a.ntheta - 1 ? tid + 1 : 0] - a.bnd32[tid]);
                const double cnt_last = (double)((int)a.N - a.bnd32[tid]);  // bnd32[ntheta] is a sentinel, not N
                a.scores[(int64_t)p * a.ntheta + tid] =
                    0.5 * (a.tmap.iv[tid] * mine - (tid == a.ntheta - 1 ? cnt_last : cnt));
            }
            if (tid == 0 && crank == 0) {
                muse_info inf;
                inf.iterations = iterations;
                inf.f_calls = f_calls;
                inf.status = (status <= MUSE_STATUS_F_CONVERGED && !isfinite(f)) ? MUSE_STATUS_NONFINITE : status;
                inf.hist_words = hist_words;
                inf.f_min = f;
                inf.gnorm = gmax;
                a.info[p] = inf;
            }
        }
        stamp(p, 7);
    }
    double last_phi;
    double score_acc[MAXB];  // per-block sums of the score terms when the solve's last pass computed them
    bool score_ready;
};

constexpr int kArgsDoubles = (int)((sizeof(BatchArgs) + 15) / 16 * 2);  // LDS copy of the kernel arguments

// The argument block is read from an LDS copy of the kernarg segment, not from the by-value
// parameter: hipcc materialises a by-value aggregate in scratch as soon as any select/phi of two
// field addresses is formed, and every access then becomes a scratch access.  LDS loads at uniform
// addresses are uniform values, so control flow on them stays scalar.
template <class Model, class Place, bool IMPLICIT = false>
__global__ void __launch_bounds__(Place::T) __attribute__((amdgpu_waves_per_eu(Place::kWavesPerEu)))
map_score_kernel(const BatchArgs /*read via the kernarg segment*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = Place::T;
    // LDS carve (all offsets multiples of 16 B): reduction scratch, L-BFGS scalars, ticket, args, x, g
    double* red = reinterpret_cast<double*>(smem);      // [2][T/64][8]
    double* shs = red + 2 * (T / 64) * 8;                // rho, gamma, alpha [3][kM]; sd [kMaxTheta]; pad
    int* ticket = reinterpret_cast<int*>(shs + 40);      // [4]
    double* args_lds = shs + 42;                         // [kArgsDoubles]
    const int tid = threadIdx.x;
    {
        typedef __attribute__((address_space(4))) const uint32_t* kernarg_ptr;
        kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t* dst = reinterpret_cast<uint32_t*>(args_lds);
        for (int w = tid; w < (int)(sizeof(BatchArgs) / 4); w += T) dst[w] = kp[w];
    }
    __syncthreads();
    const BatchArgs& a = *reinterpret_cast<const BatchArgs*>(args_lds);
    double* lds_x = args_lds + kArgsDoubles;  // [ld + 2]: elements, dummy slot (index ld), pad
    double* lds_g = lds_x + a.ld + 2;         // [ld + 2]
    if constexpr (Place::kXgLds) {
        if (tid == 0) {  // the dummy slot and the pad element (N odd) hold 0 for the kernel's lifetime
            lds_x[a.ld] = 0.0;
            lds_g[a.ld] = 0.0;
            lds_x[a.ld + 1] = 0.0;
            lds_g[a.ld + 1] = 0.0;
            if (a.N < a.ld) {
                lds_x[a.N] = 0.0;
                lds_g[a.N] = 0.0;
            }
        }
    }
    if constexpr (Place::kCluster) {
        // csize consecutive workgroups form a cluster that works on one problem at a time; problems are
        // dealt to clusters round-robin (every member computes the same sequence: no communication).
        const int csize = a.csize, cluster = blockIdx.x / csize, crank = blockIdx.x % csize;
        double* cl_scratch = a.scratch + (int64_t)cluster * a.scratch_stride;
        Solver<Model, Place> sv(a, tid, red, shs);
        sv.pack_blocks();
        sv.crank = crank;
        sv.csize = csize;
        sv.tfirst = crank * T + tid;
        sv.pstride = csize * T;
        sv.cl_counter = a.cl_counter + cluster;
        sv.cl_part = a.cl_part + (size_t)cluster * 2 * csize * 8;
        for (int p = cluster; p < a.nproblems; p += a.nclusters) {
            sv.parity = 0;
            __syncthreads();
            if constexpr (IMPLICIT) sv.run_implicit(p, cl_scratch, lds_x, lds_g);
            else sv.run(p, cl_scratch, lds_x, lds_g);
        }
    } else {
        double* wg_scratch = a.scratch + (int64_t)blockIdx.x * a.scratch_stride;
        unsigned pk0, pk1;  // the thread's packed block indices: once per kernel, not per problem
        {
            Solver<Model, Place> s0(a, tid, red, shs);
            s0.pack_blocks();
            pk0 = s0.pk[0];
            pk1 = s0.pk[1];
        }
        for (;;) {
            __syncthreads();
            if (tid == 0) ticket[0] = atomicAdd(a.work_counter, 1);
            __syncthreads();
            const int p = __builtin_amdgcn_readfirstlane(ticket[0]) - a.ticket_base;
            if (p >= a.nproblems) break;
            Solver<Model, Place> sv(a, tid, red, shs);
            sv.pk[0] = pk0;
            sv.pk[1] = pk1;
            if constexpr (IMPLICIT) sv.run_implicit(p, wg_scratch, lds_x, lds_g);
            else sv.run(p, wg_scratch, lds_x, lds_g);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Per-simulation operator kernels (API parity with the reference's per-sim interface; not the
// performance path).
template <int MODEL>
__global__ void __launch_bounds__(256) sample_kernel(BatchArgs a, uint64_t sim, double* __restrict__ x,
                                                     double* __restrict__ z) {
    const int64_t N = a.N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = a.ntheta > 1 ? block_of(a, i) : 0;
        const double sdk = a.tmap.sd[k];
        const NormalPair np = normal_pair(a.seed, sim, (uint64_t)i);
        if (MODEL == MUSE_MODEL_NOISE) {
            z[i] = np.n1;
            x[i] = np.n1 + sdk * np.n2;
        } else if (MODEL == MUSE_MODEL_FUNNEL) {
            const double zi = sdk * np.n1;
            z[i] = zi;
            x[i] = zi + np.n2;
        } else {
            z[i] = sdk * np.n1;
            x[i] = np.n2;
        }
    }
}
__global__ void __launch_bounds__(256) smooth_finish_kernel(int64_t N, const double* __restrict__ z,
                                                            const double* __restrict__ noise, double* __restrict__ x) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t im = i == 0 ? N - 1 : i - 1, ip = i == N - 1 ? 0 : i + 1;
        x[i] = fma(0.25, z[im] + z[ip], 0.5 * z[i]) + noise[i];
    }
}

// logLike and grad_z logLike (note the sign: the solver works with -logLike), plus the per-block score
// sums; one workgroup, fixed-shape reduction (same element->thread map as the solver).
template <class Model>
__global__ void __launch_bounds__(1024) loglike_kernel(BatchArgs a, const double* __restrict__ xin,
                                                       const double* __restrict__ zin, double* __restrict__ gout,
                                                       double* __restrict__ out /* [0]=logLike, [1..]=score */) {
    __shared__ double red[2 * 16 * 8];
    constexpr int T = 1024, MAXB = Model::MAXB;
    const int tid = threadIdx.x;
    int parity = 0;
    const int64_t N = a.N;
    double sum[2] = {0.0, 0.0}, mx[1] = {0.0};
    double acc[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) acc[b] = 0.0;
    const int Ni = (int)N;
    auto wrap = [&](int i) { return i < 0 ? i + Ni : (i >= Ni ? i - Ni : i); };
    // vectors are padded to the even length ld with a zero pad element (phantom zero, see for_elems)
    for_elems<T, 0, 1>(a.ld, tid, T, [&](int, int i) {
        const int k = MAXB > 1 ? block_of(a, i) : 0;
        const double ivk = a.tmap.iv[k];
        double gi;
        if constexpr (Model::kStencil) {
            const bool valid = i < Ni;
            const int ic = valid ? i : 0;
            const int im2 = wrap(ic - 2), im1 = wrap(ic - 1), ip1 = wrap(ic + 1), ip2 = wrap(ic + 2);
            const double zm2 = zin[im2], zm1 = zin[im1], z0 = zin[ic], zp1 = zin[ip1], zp2 = zin[ip2];
            const double rm = xin[im1] - fma(0.25, zm2 + z0, 0.5 * zm1);
            const double r0 = xin[ic] - fma(0.25, zm1 + zp1, 0.5 * z0);
            const double rp = xin[ip1] - fma(0.25, z0 + zp2, 0.5 * zp1);
            const double t = ivk * z0;
            sum[0] = valid ? fma(t, z0, fma(r0, r0, sum[0])) : sum[0];
            gi = valid ? t - fma(0.25, rm + rp, 0.5 * r0) : 0.0;
        } else {
            gi = Model::grad(ivk, xin[i], zin[i], sum[0]);
        }
        if (gout) gout[i] = -gi;
        const double t = Model::score_term(xin[i], zin[i]);
#pragma unroll
        for (int b = 0; b < MAXB; ++b) acc[b] += (k == b) ? t : 0.0;
    });
    block_allreduce<T, 2, 0>(sum, mx, red, parity, tid);
    block_allreduce<T, MAXB, 0>(acc, mx, red, parity, tid);
    if (tid == 0) {
        out[0] = -(0.5 * (sum[0] + a.f_const));
        for (int b = 0; b < MAXB; ++b)
            if (b < a.ntheta) out[1 + b] = 0.5 * (a.tmap.iv[b] * acc[b] - (double)(a.bnd[b + 1] - a.bnd[b]));
    }
}

}  // namespace muse

// ================================================================================================
// Host side: context, workspace, launches, C ABI.
// ================================================================================================
using namespace muse;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(MUSE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

struct muse_ctx {
    int model = 0, ntheta = 1, device = 0, placement = -1, num_cus = 0;
    int64_t N = 0, ld = 0;
    int64_t bnd[kMaxTheta + 1] = {0};
    hipStream_t stream = nullptr, own_stream = nullptr;
    double* x_data = nullptr;
    bool has_data = false;
    double* zhat = nullptr;
    int64_t zhat_slots = 0;
    double* scratch = nullptr;
    size_t scratch_doubles = 0;
    int* counter = nullptr;
    double* tmp = nullptr;  // 3 vectors for the per-sim operator entry points
    ThetaSet* tsample_dev = nullptr;
    ThetaSet* tsample_pin = nullptr;
    // result areas: device + pinned host, each [cap] scores and infos
    double* scores_dev[kResultAreas] = {nullptr};
    muse_info* info_dev[kResultAreas] = {nullptr};
    double* scores_pin[kResultAreas] = {nullptr};
    muse_info* info_pin[kResultAreas] = {nullptr};
    int64_t res_cap[kResultAreas] = {0};
    int64_t res_n[kResultAreas] = {0};
    double* small_dev = nullptr;  // 16 doubles
    hipEvent_t ev0 = nullptr, ev1 = nullptr, last0 = nullptr, last1 = nullptr;
    bool ev_valid = false;
    hipEvent_t area_done[kResultAreas] = {nullptr};  // recorded after an area's device->host copies
    // live kernel timing: a ring of event pairs, one per solver launch (muse_profile_*)
    std::vector<hipEvent_t> prof_ev;
    int prof_count = 0;
    bool prof_on = false;
    double* ncache = nullptr;            // normals cache [ncache_slots][2][ld] (muse_run, FD batches)
    int64_t ncache_slots = 0;
    unsigned int* cl_counter = nullptr;  // cluster mode: [cl_cap] arrival counters
    double* cl_part = nullptr;           // [cl_cap][2][kMaxCluster][8]
    int cl_cap = 0;
    int* error_flag = nullptr;           // pinned, device-mapped
    int debug = 0;
    unsigned int ticket_base = 0;  // value of the device ticket counter when the next launch starts
    bool timing = false;           // record an event pair around every solver launch (muse_set_timing; costs ~12 us per launch)
    unsigned long long* stamps = nullptr;
    int64_t stamps_cap = 0;
    void* comm = nullptr;  // ncclComm_t (muse_comm.cpp)
    double* comm_buf = nullptr;
    size_t comm_buf_doubles = 0;
};

static void make_thetaset(const muse_ctx* c, const double* theta, ThetaSet& t) {
    memset(&t, 0, sizeof(t));
    for (int k = 0; k < c->ntheta; ++k) {
        t.theta[k] = theta[k];
        t.sd[k] = exp(0.5 * theta[k]);
        t.iv[k] = exp(-theta[k]);
    }
}
static double theta_const(const muse_ctx* c, const double* theta) {
    double cst = 0.0;
    for (int k = 0; k < c->ntheta; ++k) cst += (double)(c->bnd[k + 1] - c->bnd[k]) * theta[k];
    return cst;
}

#ifndef MUSE_STENCIL_U
#define MUSE_STENCIL_U 4
#endif
constexpr int kStencilU = MUSE_STENCIL_U;  // pairs per trip for the stencil model (register budget: see tools/regs.py)

enum PlaceId { P_S256 = 0, P_S512 = 1, P_R256x1 = 2, P_R512x4 = 3, P_R512x10 = 4, P_C256 = 5 };

// Cluster size: a function of N alone (results must not depend on how many problems share a launch).
static int cluster_size(int64_t N) {
    if (const char* e = getenv("MUSE_DEBUG_CLUSTER_SIZE")) return atoi(e);  // tuning aid
    return N >= 4194304 ? 16 : (N >= kClusterMinN ? 4 : 1);
}

// Workgroup size is a function of N alone (256 threads for N <= 512, else 512), so that the
// streaming and the resident policy reduce in the same order and give bitwise equal results.
static int choose_place(const muse_ctx* c) {
    const bool small = c->N <= 512;
    // cluster mode: for the stencil model the neighbours owned by other workgroups become visible through
    // the agent-scope release/acquire of the cluster reduction that ends every pass (pass_barrier where a
    // pass has no reduction)
    if (c->N >= kClusterMinN) return P_C256;
    if (c->model == MUSE_MODEL_SMOOTH || c->placement == 0 || c->N > kMaxResidentN) return small ? P_S256 : P_S512;
    if (small) return P_R256x1;
    if (c->N <= 4096) return P_R512x4;
    return P_R512x10;
}
static bool ncache_applies(const muse_ctx* c) {
    static const bool off = getenv("MUSE_DEBUG_NO_NCACHE") != nullptr;  // tuning aid
    return !off && choose_place(c) == P_R512x10;
}
static int place_threads(int pl) { return (pl == P_S256 || pl == P_R256x1 || pl == P_C256) ? 256 : 512; }
static int place_wgs_per_cu(int pl) { return (pl == P_S256 || pl == P_R256x1) ? 4 : ((pl == P_R512x10) ? 1 : 2); }
static size_t place_lds(const muse_ctx* c, int pl) {
    size_t fixed = (size_t)(2 * (place_threads(pl) / 64) * 8 + 42 + kArgsDoubles) * sizeof(double);
    if (pl == P_R512x10) fixed += (size_t)2 * (c->ld + 2) * sizeof(double);
    return fixed;
}
static int64_t place_scratch_vectors(int pl) { return (pl == P_S256 || pl == P_S512 || pl == P_C256) ? 4 + 2 * kM + 1 : 2 * kM; }

static int ensure_zhat(muse_ctx* c, int64_t slots) {
    if (slots <= c->zhat_slots) return MUSE_OK;
    double* nz = nullptr;
    if (hipMalloc(&nz, (size_t)slots * c->ld * sizeof(double)) != hipSuccess)
        return fail(MUSE_ERR_ALLOC, "hipMalloc(zhat) failed");
    HIPCHK(hipMemsetAsync(nz, 0, (size_t)slots * c->ld * sizeof(double), c->stream));
    if (c->zhat) {
        HIPCHK(hipMemcpyAsync(nz, c->zhat, (size_t)c->zhat_slots * c->ld * sizeof(double), hipMemcpyDeviceToDevice,
                              c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(c->zhat));
    }
    c->zhat = nz;
    c->zhat_slots = slots;
    return MUSE_OK;
}
static int ensure_scratch(muse_ctx* c, size_t doubles) {
    if (doubles <= c->scratch_doubles) return MUSE_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->scratch) HIPCHK(hipFree(c->scratch));
    c->scratch = nullptr;
    c->scratch_doubles = 0;
    if (hipMalloc(&c->scratch, doubles * sizeof(double)) != hipSuccess)
        return fail(MUSE_ERR_ALLOC, "hipMalloc(scratch) failed");
    c->scratch_doubles = doubles;
    return MUSE_OK;
}
// The normals cache exists only where the sampler is a large share of a problem and the placement supports it
// (the LDS-resident layout, 4096 < N <= kMaxResidentN); a failed allocation just means no caching.
static bool ncache_applies(const muse_ctx* c);
static bool ensure_ncache(muse_ctx* c, int64_t slots) {
    if (!ncache_applies(c)) return false;
    if (slots <= c->ncache_slots) return true;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
    if (c->ncache) hipFree(c->ncache);
    c->ncache = nullptr;
    c->ncache_slots = 0;
    if (hipMalloc(&c->ncache, (size_t)slots * 2 * c->ld * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    c->ncache_slots = slots;
    return true;
}
// One pinned, device-mapped host block per result area: [cap*ntheta] scores then [cap] infos.  The
// solver kernel writes an element's score and info straight into it (a few hundred posted PCIe
// writes per batch), so no device->host copy sits between consecutive launches.
static size_t result_bytes(const muse_ctx* c, int64_t cap) {
    return (size_t)cap * (size_t)c->ntheta * sizeof(double) + (size_t)cap * sizeof(muse_info);
}
static int ensure_results(muse_ctx* c, int area, int64_t n) {
    if (n <= c->res_cap[area]) return MUSE_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->scores_pin[area]) HIPCHK(hipHostFree(c->scores_pin[area]));
    const int64_t cap = n + n / 2 + 16;
    HIPCHK(hipHostMalloc(&c->scores_pin[area], result_bytes(c, cap), hipHostMallocDefault));
    c->info_pin[area] = reinterpret_cast<muse_info*>(c->scores_pin[area] + cap * c->ntheta);
    c->scores_dev[area] = c->scores_pin[area];  // unified addressing: the kernel stores through the same pointers
    c->info_dev[area] = c->info_pin[area];
    c->res_cap[area] = cap;
    return MUSE_OK;
}

template <class Model, class Place, bool IMPLICIT = false>
static int launch_one(muse_ctx* c, const BatchArgs& a, int grid, size_t lds) {
    auto kern = map_score_kernel<Model, Place, IMPLICIT>;
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Place::T), lds, c->stream, a);
    HIPCHK(hipGetLastError());
    return MUSE_OK;
}
template <class Model>
static int launch_place(muse_ctx* c, const BatchArgs& a, int pl, int grid, size_t lds) {
    if constexpr (Model::kStencil) {
        if (pl == P_C256) return launch_one<Model, PlaceStreaming<256, true, kStencilU>>(c, a, grid, lds);
        if (pl == P_S256) return launch_one<Model, PlaceStreaming<256, false, kStencilU>>(c, a, grid, lds);
        return launch_one<Model, PlaceStreaming<512, false, kStencilU>>(c, a, grid, lds);
    } else {
        switch (pl) {
            case P_R256x1: return launch_one<Model, PlaceResident<256, 1, false>>(c, a, grid, lds);
            case P_R512x4: return launch_one<Model, PlaceResident<512, 4, false>>(c, a, grid, lds);
            case P_R512x10: return launch_one<Model, PlaceResident<512, 10, true>>(c, a, grid, lds);
            case P_C256: return launch_one<Model, PlaceStreaming<256, true>>(c, a, grid, lds);
            case P_S256: return launch_one<Model, PlaceStreaming<256>>(c, a, grid, lds);
            default: return launch_one<Model, PlaceStreaming<512>>(c, a, grid, lds);
        }
    }
}

// The implicit-differentiation H runs in the streaming policy only (single workgroup, or a cluster for large N).
template <class Model>
static int launch_place_implicit(muse_ctx* c, const BatchArgs& a, int pl, int grid, size_t lds) {
    constexpr int U = Model::kStencil ? kStencilU : 4;
    if (pl == P_C256) return launch_one<Model, PlaceStreaming<256, true, U>, true>(c, a, grid, lds);
    return launch_one<Model, PlaceStreaming<512, false, U>, true>(c, a, grid, lds);
}

// Fill the common fields and launch the solver for `a.nproblems` elements.
static int launch_batch(muse_ctx* c, BatchArgs& a) {
    a.N = c->N;
    a.ld = c->ld;
    a.ntheta = c->ntheta;
    for (int k = 0; k <= kMaxTheta; ++k) {
        a.bnd[k] = c->bnd[k];
        a.bnd32[k] = k < c->ntheta ? (int)c->bnd[k] : 0x7fffffff;
    }
    a.x_data = c->x_data;
    a.zhat = c->zhat;
    a.work_counter = c->counter;
    a.debug = c->debug;
    a.stamps = (c->stamps && a.nproblems <= c->stamps_cap) ? c->stamps : nullptr;
    const bool implicit = a.kind == BATCH_IMPLICIT;
    const int pl = implicit ? (c->N >= kClusterMinN ? P_C256 : P_S512) : choose_place(c);
    int grid = c->num_cus * place_wgs_per_cu(pl);
    a.csize = 1;
    a.nclusters = 0;
    if (pl == P_C256) {
        // every workgroup of a cluster must be resident at once (they wait for each other): size the grid
        // from 2 workgroups of 256 threads per CU, which the kernel's register budget always admits
        a.csize = cluster_size(c->N);
        int ncl = grid / a.csize;
        if (ncl > a.nproblems) ncl = a.nproblems;
        if (ncl < 1) ncl = 1;
        a.nclusters = ncl;
        grid = ncl * a.csize;
        if (ncl > c->cl_cap) {
            HIPCHK(hipStreamSynchronize(c->stream));
            if (c->cl_counter) HIPCHK(hipFree(c->cl_counter));
            if (c->cl_part) HIPCHK(hipFree(c->cl_part));
            HIPCHK(hipMalloc(&c->cl_counter, (size_t)ncl * sizeof(unsigned int)));
            HIPCHK(hipMalloc(&c->cl_part, (size_t)ncl * 2 * kMaxCluster * 8 * sizeof(double)));
            c->cl_cap = ncl;
        }
        HIPCHK(hipMemsetAsync(c->cl_counter, 0, (size_t)ncl * sizeof(unsigned int), c->stream));
        a.cl_counter = c->cl_counter;
        a.cl_part = c->cl_part;
    } else {
        if (grid > a.nproblems) grid = a.nproblems;
        if (grid < 1) grid = 1;
    }
    a.error_flag = c->error_flag;
    a.scratch_stride = place_scratch_vectors(pl) * c->ld;
    int rc = ensure_scratch(c, (size_t)(pl == P_C256 ? a.nclusters : grid) * a.scratch_stride);
    if (rc) return rc;
    a.scratch = c->scratch;
    const size_t lds = place_lds(c, pl);
    // Tickets: every workgroup draws tickets until it draws one past the batch, so a launch advances
    // the counter by exactly nproblems + grid -- no per-launch memset.  Wrap-around: reset explicitly.
    if (c->ticket_base > 0x70000000u) {
        HIPCHK(hipMemsetAsync(c->counter, 0, 16, c->stream));
        c->ticket_base = 0;
    }
    a.ticket_base = (int)c->ticket_base;
    c->ticket_base += (unsigned)a.nproblems + (unsigned)grid;
    hipEvent_t e0 = c->ev0, e1 = c->ev1;
    if (c->prof_on && (size_t)(2 * c->prof_count + 1) < c->prof_ev.size()) {
        e0 = c->prof_ev[2 * c->prof_count];
        e1 = c->prof_ev[2 * c->prof_count + 1];
        c->prof_count += 1;
    }
    const bool timed = c->timing || c->prof_on;
    if (timed) HIPCHK(hipEventRecord(e0, c->stream));
    if (implicit) {
        if (c->model == MUSE_MODEL_NOISE) rc = launch_place_implicit<NoiseModel>(c, a, pl, grid, lds);
        else if (c->model == MUSE_MODEL_FUNNEL)
            rc = c->ntheta == 1   ? launch_place_implicit<FunnelModel<1>>(c, a, pl, grid, lds)
                 : c->ntheta == 2 ? launch_place_implicit<FunnelModel<2>>(c, a, pl, grid, lds)
                 : c->ntheta <= 4 ? launch_place_implicit<FunnelModel<4>>(c, a, pl, grid, lds)
                                  : launch_place_implicit<FunnelModel<kMaxTheta>>(c, a, pl, grid, lds);
        else
            rc = c->ntheta <= 2   ? launch_place_implicit<SmoothModel<2>>(c, a, pl, grid, lds)
                 : c->ntheta <= 4 ? launch_place_implicit<SmoothModel<4>>(c, a, pl, grid, lds)
                                  : launch_place_implicit<SmoothModel<kMaxTheta>>(c, a, pl, grid, lds);
    } else if (c->model == MUSE_MODEL_NOISE) rc = launch_place<NoiseModel>(c, a, pl, grid, lds);
    else if (c->model == MUSE_MODEL_FUNNEL)
        rc = c->ntheta == 1   ? launch_place<FunnelModel<1>>(c, a, pl, grid, lds)
             : c->ntheta == 2 ? launch_place<FunnelModel<2>>(c, a, pl, grid, lds)
             : c->ntheta <= 4 ? launch_place<FunnelModel<4>>(c, a, pl, grid, lds)
                              : launch_place<FunnelModel<kMaxTheta>>(c, a, pl, grid, lds);
    else
        rc = c->ntheta <= 2   ? launch_place<SmoothModel<2>>(c, a, pl, grid, lds)
             : c->ntheta <= 4 ? launch_place<SmoothModel<4>>(c, a, pl, grid, lds)
                              : launch_place<SmoothModel<kMaxTheta>>(c, a, pl, grid, lds);
    if (rc) return rc;
    if (timed) {
        HIPCHK(hipEventRecord(e1, c->stream));
        c->last0 = e0;
        c->last1 = e1;
        c->ev_valid = true;
    }
    return MUSE_OK;
}

extern "C" {

const char* muse_last_error(void) { return g_err.c_str(); }
int64_t muse_max_resident_n(void) { return kMaxResidentN; }

int muse_ctx_create(int model, int64_t N, int ntheta, int device, muse_ctx** out) {
    if (!out) return fail(MUSE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (model < 0 || model > 2) return fail(MUSE_ERR_INVALID, "unknown model");
    if (N < 1) return fail(MUSE_ERR_INVALID, "N must be >= 1");
    if (ntheta < 1 || ntheta > kMaxTheta) return fail(MUSE_ERR_INVALID, "ntheta must be in [1, MUSE_MAX_THETA]");
    if (model == MUSE_MODEL_NOISE && ntheta != 1) return fail(MUSE_ERR_INVALID, "MUSE_MODEL_NOISE has ntheta = 1");
    if (ntheta > N) return fail(MUSE_ERR_INVALID, "ntheta must be <= N");
    if (model == MUSE_MODEL_SMOOTH && N < 5) return fail(MUSE_ERR_INVALID, "MUSE_MODEL_SMOOTH needs N >= 5");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MUSE_ERR_HIP, "no HIP device available (libmuse_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MUSE_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    muse_ctx* c = new muse_ctx();
    c->model = model;
    c->N = N;
    c->ld = (N + 1) & ~(int64_t)1;
    c->ntheta = ntheta;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    for (int k = 0; k <= kMaxTheta; ++k) {
        const int kk = k < ntheta ? k : ntheta;
        c->bnd[k] = ((int64_t)kk * N + ntheta - 1) / ntheta;
    }
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(hipMalloc(&c->x_data, (size_t)c->ld * sizeof(double)));
    HIPCHK(hipMalloc(&c->counter, 16));
    HIPCHK(hipMemset(c->counter, 0, 16));
    HIPCHK(hipHostMalloc(&c->error_flag, 64, hipHostMallocDefault));
    *c->error_flag = 0;
    HIPCHK(hipMalloc(&c->tmp, (size_t)3 * c->ld * sizeof(double)));
    HIPCHK(hipMalloc(&c->small_dev, 16 * sizeof(double)));
    HIPCHK(hipMalloc(&c->tsample_dev, 2 * kMaxTheta * sizeof(ThetaSet)));
    HIPCHK(hipHostMalloc(&c->tsample_pin, 2 * kMaxTheta * sizeof(ThetaSet), hipHostMallocDefault));
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    for (int r = 0; r < kResultAreas; ++r) HIPCHK(hipEventCreateWithFlags(&c->area_done[r], hipEventDisableTiming));
    HIPCHK(hipMemsetAsync(c->x_data, 0, (size_t)c->ld * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->tmp, 0, (size_t)3 * c->ld * sizeof(double), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = c;
    return MUSE_OK;
}

int muse_comm_destroy(muse_ctx* ctx);

// accessors for muse_comm.cpp (the context layout is private to this file)
int muse_set_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }
int muse_ctx_comm_slot(muse_ctx* c, void*** comm, int* device, void** stream) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    *comm = &c->comm;
    *device = c->device;
    *stream = (void*)c->stream;
    return MUSE_OK;
}
int muse_ctx_area_event(muse_ctx* c, int area, void** event, int* ntheta) {
    if (!c || area < 0 || area >= kResultAreas) return fail(MUSE_ERR_INVALID, "bad result_area");
    *event = (void*)c->area_done[area];
    *ntheta = c->ntheta;
    return MUSE_OK;
}
int muse_ctx_comm_buffer(muse_ctx* c, size_t doubles, double** buf) {
    if (doubles > c->comm_buf_doubles) {
        if (c->comm_buf) HIPCHK(hipFree(c->comm_buf));
        c->comm_buf = nullptr;
        c->comm_buf_doubles = 0;
        if (hipMalloc(&c->comm_buf, doubles * sizeof(double)) != hipSuccess)
            return fail(MUSE_ERR_ALLOC, "hipMalloc(comm buffer) failed");
        c->comm_buf_doubles = doubles;
    }
    *buf = c->comm_buf;
    return MUSE_OK;
}

int muse_ctx_destroy(muse_ctx* c) {
    if (!c) return MUSE_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    muse_comm_destroy(c);
    hipFree(c->cl_counter); hipFree(c->cl_part); hipHostFree(c->error_flag); hipFree(c->ncache);
    hipFree(c->x_data); hipFree(c->zhat); hipFree(c->scratch); hipFree(c->counter); hipFree(c->tmp);
    hipFree(c->small_dev); hipFree(c->tsample_dev); hipHostFree(c->tsample_pin);
    if (c->comm_buf) hipFree(c->comm_buf);
    for (int r = 0; r < kResultAreas; ++r) {
        hipHostFree(c->scores_pin[r]);
    }
    hipEventDestroy(c->ev0); hipEventDestroy(c->ev1);
    for (int r = 0; r < kResultAreas; ++r) hipEventDestroy(c->area_done[r]);
    for (hipEvent_t e : c->prof_ev) hipEventDestroy(e);
    hipStreamDestroy(c->own_stream);
    delete c;
    return MUSE_OK;
}

static int check_ctx(muse_ctx* c) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    return MUSE_OK;
}
static hipMemcpyKind in_kind(int mem) { return mem == MUSE_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice; }
static hipMemcpyKind out_kind(int mem) { return mem == MUSE_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost; }

int muse_set_data(muse_ctx* c, const double* x, int mem) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!x) return fail(MUSE_ERR_INVALID, "x is NULL");
    HIPCHK(hipMemcpyAsync(c->x_data, x, (size_t)c->N * sizeof(double), in_kind(mem), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->has_data = true;
    return MUSE_OK;
}
int muse_set_stream(muse_ctx* c, void* s) {
    int rc = check_ctx(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return MUSE_OK;
}
int muse_set_placement(muse_ctx* c, int placement) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    if (placement < -1 || placement > 1) return fail(MUSE_ERR_INVALID, "placement must be -1, 0 or 1");
    if (placement == 1 && (c->N > kMaxResidentN || c->model == MUSE_MODEL_SMOOTH))
        return fail(MUSE_ERR_INVALID, "resident placement not available for this problem");
    c->placement = placement;
    return MUSE_OK;
}
int muse_synchronize(muse_ctx* c) {
    int rc = check_ctx(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return MUSE_OK;
}
int muse_last_kernel_ms(muse_ctx* c, float* ms) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!ms) return fail(MUSE_ERR_INVALID, "ms is NULL");
    if (!c->ev_valid) return fail(MUSE_ERR_INVALID, "no solver launch recorded yet");
    HIPCHK(hipEventSynchronize(c->last1));
    HIPCHK(hipEventElapsedTime(ms, c->last0, c->last1));
    return MUSE_OK;
}

int muse_debug_stamps(muse_ctx* c, int64_t nproblems, unsigned long long* out) {  // diagnostic aid
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!out) {  // arm
        if (c->stamps) HIPCHK(hipFree(c->stamps));
        HIPCHK(hipMalloc(&c->stamps, (size_t)nproblems * 16 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(c->stamps, 0, (size_t)nproblems * 16 * sizeof(unsigned long long)));
        c->stamps_cap = nproblems;
        return MUSE_OK;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->stamps, (size_t)nproblems * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MUSE_OK;
}
int muse_set_timing(muse_ctx* c, int enabled) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    c->timing = enabled != 0;
    return MUSE_OK;
}
int muse_debug_flags(muse_ctx* c, int flags) {  // not part of the public header: profiling aid
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    c->debug = flags;
    return MUSE_OK;
}
int muse_profile_begin(muse_ctx* c, int max_launches) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (max_launches < 1 || max_launches > 65536) return fail(MUSE_ERR_INVALID, "max_launches out of range");
    HIPCHK(hipStreamSynchronize(c->stream));
    while (c->prof_ev.size() < (size_t)(2 * max_launches + 2)) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->prof_ev.push_back(e);
    }
    c->prof_count = 0;
    c->prof_on = true;
    return MUSE_OK;
}
int muse_profile_end(muse_ctx* c, float* ms_out, int cap, int* count) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!count) return fail(MUSE_ERR_INVALID, "count is NULL");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->prof_on = false;
    *count = c->prof_count;
    for (int k = 0; k < c->prof_count && k < cap && ms_out; ++k)
        HIPCHK(hipEventElapsedTime(&ms_out[k], c->prof_ev[2 * k], c->prof_ev[2 * k + 1]));
    return MUSE_OK;
}

static void base_args(muse_ctx* c, BatchArgs& a, const double* theta) {
    memset(&a, 0, sizeof(a));
    a.N = c->N;
    a.ld = c->ld;
    a.ntheta = c->ntheta;
    for (int k = 0; k <= kMaxTheta; ++k) {
        a.bnd[k] = c->bnd[k];
        a.bnd32[k] = k < c->ntheta ? (int)c->bnd[k] : 0x7fffffff;
    }
    make_thetaset(c, theta, a.tmap);
    a.f_const = theta_const(c, theta);
    a.fid_slot = -1;
    a.nstd = 0x7fffffff;  // no normals-only elements
}

int muse_sample_x_z(muse_ctx* c, uint64_t seed, int64_t sim, const double* theta, double* x_out, double* z_out,
                    int mem) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta || (!x_out && !z_out)) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (sim < 0) return fail(MUSE_ERR_INVALID, "sim must be >= 0");
    BatchArgs a;
    base_args(c, a, theta);
    a.seed = seed;
    double *dx = c->tmp, *dz = c->tmp + c->ld, *dn = c->tmp + 2 * c->ld;
    const int grid = (int)((c->N + 255) / 256 < 4096 ? (c->N + 255) / 256 : 4096);
    if (c->model == MUSE_MODEL_NOISE) hipLaunchKernelGGL(sample_kernel<MUSE_MODEL_NOISE>, dim3(grid), dim3(256), 0, c->stream, a, (uint64_t)sim, dx, dz);
    else if (c->model == MUSE_MODEL_FUNNEL) hipLaunchKernelGGL(sample_kernel<MUSE_MODEL_FUNNEL>, dim3(grid), dim3(256), 0, c->stream, a, (uint64_t)sim, dx, dz);
    else {
        hipLaunchKernelGGL(sample_kernel<MUSE_MODEL_SMOOTH>, dim3(grid), dim3(256), 0, c->stream, a, (uint64_t)sim, dn, dz);
        hipLaunchKernelGGL(smooth_finish_kernel, dim3(grid), dim3(256), 0, c->stream, c->N, dz, dn, dx);
    }
    HIPCHK(hipGetLastError());
    if (x_out) HIPCHK(hipMemcpyAsync(x_out, dx, (size_t)c->N * sizeof(double), out_kind(mem), c->stream));
    if (z_out) HIPCHK(hipMemcpyAsync(z_out, dz, (size_t)c->N * sizeof(double), out_kind(mem), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MUSE_OK;
}

static int run_loglike(muse_ctx* c, const double* x, const double* z, const double* theta, double* gdev, int mem) {
    BatchArgs a;
    base_args(c, a, theta);
    double *dx = c->tmp, *dz = c->tmp + c->ld;
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)c->N * sizeof(double), in_kind(mem), c->stream));
    HIPCHK(hipMemcpyAsync(dz, z, (size_t)c->N * sizeof(double), in_kind(mem), c->stream));
    if (c->model == MUSE_MODEL_NOISE) hipLaunchKernelGGL(loglike_kernel<NoiseModel>, dim3(1), dim3(1024), 0, c->stream, a, dx, dz, gdev, c->small_dev);
    else if (c->model == MUSE_MODEL_FUNNEL) hipLaunchKernelGGL(loglike_kernel<FunnelModel<kMaxTheta>>, dim3(1), dim3(1024), 0, c->stream, a, dx, dz, gdev, c->small_dev);
    else hipLaunchKernelGGL(loglike_kernel<SmoothModel<kMaxTheta>>, dim3(1), dim3(1024), 0, c->stream, a, dx, dz, gdev, c->small_dev);
    HIPCHK(hipGetLastError());
    return MUSE_OK;
}

int muse_logLike_and_grad_z(muse_ctx* c, const double* x, const double* z, const double* theta, double* logLike_out,
                            double* grad_out, int mem) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!x || !z || !theta) return fail(MUSE_ERR_INVALID, "NULL argument");
    double* gdev = c->tmp + 2 * c->ld;
    rc = run_loglike(c, x, z, theta, grad_out ? gdev : nullptr, mem);
    if (rc) return rc;
    double small[1 + kMaxTheta];
    HIPCHK(hipMemcpyAsync(small, c->small_dev, sizeof(small), hipMemcpyDeviceToHost, c->stream));
    if (grad_out) HIPCHK(hipMemcpyAsync(grad_out, gdev, (size_t)c->N * sizeof(double), out_kind(mem), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (logLike_out) *logLike_out = small[0];
    return MUSE_OK;
}

int muse_grad_theta(muse_ctx* c, const double* x, const double* z, const double* theta, double* g_out, int mem) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!x || !z || !theta || !g_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    rc = run_loglike(c, x, z, theta, nullptr, mem);
    if (rc) return rc;
    double small[1 + kMaxTheta];
    HIPCHK(hipMemcpyAsync(small, c->small_dev, sizeof(small), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->ntheta; ++k) g_out[k] = small[1 + k];
    return MUSE_OK;
}

static int enqueue_results_copy(muse_ctx* c, int area, int64_t n) {
    // results are already on their way to pinned host memory; mark the point at which they are complete
    c->res_n[area] = n;
    HIPCHK(hipEventRecord(c->area_done[area], c->stream));
    return MUSE_OK;
}

int muse_zhat_at_theta(muse_ctx* c, const double* x, const double* z0, const double* theta, double atol, double* z_out,
                       muse_info* info, int mem) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!x || !z0 || !theta || !z_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    // the single-element solve uses a private zhat slot after the batch slots
    rc = ensure_zhat(c, c->zhat_slots > 0 ? c->zhat_slots : 1);
    if (rc) return rc;
    rc = ensure_results(c, kResultAreas - 1, 1);
    if (rc) return rc;
    // stage z0 into a scratch slot: reuse tmp[1] as the z vector via a one-slot zhat view
    BatchArgs a;
    base_args(c, a, theta);
    a.kind = BATCH_SINGLE;
    a.atol = atol;
    a.nproblems = 1;
    a.store_zhat = 1;
    a.slot0 = 0;
    double* dx = c->tmp;
    double* dz = c->tmp + c->ld;
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)c->N * sizeof(double), in_kind(mem), c->stream));
    HIPCHK(hipMemcpyAsync(dz, z0, (size_t)c->N * sizeof(double), in_kind(mem), c->stream));
    a.x_given = dx;
    a.scores = c->scores_dev[kResultAreas - 1];
    a.info = c->info_dev[kResultAreas - 1];
    // run with zhat pointing at the tmp z vector (slot 0 of a 1-slot view)
    double* saved = c->zhat;
    c->zhat = dz;
    rc = launch_batch(c, a);
    c->zhat = saved;
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(z_out, dz, (size_t)c->N * sizeof(double), out_kind(mem), c->stream));
    rc = enqueue_results_copy(c, kResultAreas - 1, 1);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (info) *info = c->info_pin[kResultAreas - 1][0];
    return MUSE_OK;
}

// The batched map with the scores directed at `scores_dev` (any device-accessible buffer of n*ntheta
// doubles; NULL = the area's pinned host block).  muse_comm.cpp points it at the send buffer of the
// RCCL all-gather so that the scores never visit the host between the solver and the collective.
static int map_async_impl(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                          const double* theta, double atol, int z0_mode, int area, double* scores_dev, int ncache_mode);
int muse_internal_map_async(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                            const double* theta, double atol, int z0_mode, int area, double* scores_dev) {
    return map_async_impl(c, seed, sim_begin, sim_end, include_data, theta, atol, z0_mode, area, scores_dev, 0);
}
// ncache_mode 1: the batch also stores the normals of its simulations; 2: it loads them (same seed and range as the
// storing batch of the same host call).  Silently 0 where the cache does not apply.
static int map_async_impl(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                          const double* theta, double atol, int z0_mode, int area, double* scores_dev, int ncache_mode) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta) return fail(MUSE_ERR_INVALID, "theta is NULL");
    if (sim_end < sim_begin || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad sim range");
    if (z0_mode < MUSE_Z0_ZERO || z0_mode > MUSE_Z0_WARM) return fail(MUSE_ERR_INVALID, "bad z0_mode");
    if (area < 0 || area >= kResultAreas) return fail(MUSE_ERR_INVALID, "bad result_area");
    if (include_data && !c->has_data) return fail(MUSE_ERR_NODATA, "include_data set but muse_set_data was not called");
    const int64_t n = (sim_end - sim_begin) + (include_data ? 1 : 0);
    if (n == 0) { c->res_n[area] = 0; return MUSE_OK; }
    if (n > 0x7fffffff) return fail(MUSE_ERR_INVALID, "batch too large");
    rc = ensure_zhat(c, n);
    if (rc) return rc;
    rc = ensure_results(c, area, n);
    if (rc) return rc;
    BatchArgs a;
    base_args(c, a, theta);
    a.kind = BATCH_STD;
    a.seed = seed;
    a.atol = atol;
    a.nproblems = (int)n;
    a.include_data = include_data ? 1 : 0;
    a.z0_mode = z0_mode;
    a.store_zhat = 1;
    a.sim_begin = sim_begin;
    a.slot0 = 0;
    a.scores = scores_dev ? scores_dev : c->scores_dev[area];
    a.info = c->info_dev[area];
    if (ncache_mode != 0 && sim_end > sim_begin && ensure_ncache(c, sim_end - sim_begin)) {
        a.ncache = c->ncache;
        a.ncache_sim0 = sim_begin;
        a.ncache_count = (int)(sim_end - sim_begin);
        a.ncache_mode = ncache_mode;
    }
    rc = launch_batch(c, a);
    if (rc) return rc;
    return enqueue_results_copy(c, area, n);
}

int muse_map_and_score_batch_async(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                                   const double* theta, double atol, int z0_mode, int area) {
    return muse_internal_map_async(c, seed, sim_begin, sim_end, include_data, theta, atol, z0_mode, area, nullptr);
}

// Wait for an event by polling its signal (hipEventQuery) before falling back to the runtime's blocking
// wait: the pipelined host loop waits ~50 us at a time, and the runtime's own wait was measured to fall
// into a mode with ~0.4 ms wake-ups for stretches of a hundred launches (8x the step time).
int muse_wait_event(void* event) {
    hipEvent_t ev = (hipEvent_t)event;
    for (int spin = 0; spin < 4000000; ++spin) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return MUSE_OK;
        if (e != hipErrorNotReady) return fail(MUSE_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
        __builtin_ia32_pause();
    }
    HIPCHK(hipEventSynchronize(ev));
    return MUSE_OK;
}

int muse_batch_wait(muse_ctx* c, int area, double* g_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (area < 0 || area >= kResultAreas) return fail(MUSE_ERR_INVALID, "bad result_area");
    rc = muse_wait_event(c->area_done[area]);  // this area only: later launches keep running
    if (rc) return rc;
    if (*c->error_flag) {
        *c->error_flag = 0;
        return fail(MUSE_ERR_HIP, "a cluster wait expired inside the solver kernel (workgroups of a cluster were not co-resident)");
    }
    const int64_t n = c->res_n[area];
    if (g_out && n) memcpy(g_out, c->scores_pin[area], (size_t)n * c->ntheta * sizeof(double));
    if (info_out && n) memcpy(info_out, c->info_pin[area], (size_t)n * sizeof(muse_info));
    return MUSE_OK;
}

int muse_map_and_score_batch(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                             const double* theta, double atol, int z0_mode, double* g_out, muse_info* info_out) {
    int rc = muse_map_and_score_batch_async(c, seed, sim_begin, sim_end, include_data, theta, atol, z0_mode, 0);
    if (rc) return rc;
    return muse_batch_wait(c, 0, g_out, info_out);
}

// ---- the muse! outer loop in native host code (see muse_hip.h) ----------------------------------------
// inverse of a small dense matrix (n <= MUSE_MAX_THETA) by Gauss-Jordan with partial pivoting; false if singular
static bool small_inverse(int n, const double* A, double* inv) {
    double M[kMaxTheta][2 * kMaxTheta];
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
            for (int j = 0; j < 2 * n; ++j) std::swap(M[piv][j], M[col][j]);
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

int muse_run(muse_ctx* c, uint64_t seed, const double* theta0, const muse_run_options* o, int32_t* niter_out,
             double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !o || !niter_out || !theta_out || !hist_out || !gsims_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (o->nsims < 2 || o->maxsteps < 1) return fail(MUSE_ERR_INVALID, "muse_run needs nsims >= 2 and maxsteps >= 1");
    if (o->prior_kind != 0 && o->prior_kind != 1) return fail(MUSE_ERR_INVALID, "prior_kind must be 0 (flat) or 1 (Gaussian)");
    if (!c->has_data) return fail(MUSE_ERR_NODATA, "muse_run needs the observed data (muse_set_data)");
    const int nt = c->ntheta, S = o->nsims;
    const int64_t H = MUSE_RUN_HIST(nt);
    double theta[kMaxTheta], gprior[kMaxTheta], hprior[kMaxTheta];
    for (int k = 0; k < nt; ++k) theta[k] = theta0[k];
    std::vector<double> g((size_t)(S + 1) * nt);
    std::vector<muse_info> info((size_t)S + 1);
    int n = 0;
    for (int i = 1; i <= o->maxsteps; ++i) {
        const double t_start = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                   std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-9;
        if (i > 2) {  // convergence on the last two records (src/muse.jl:163-166)
            const double* h1 = hist_out + (int64_t)(i - 2) * H;  // record i-1
            const double* h0 = hist_out + (int64_t)(i - 3) * H;  // record i-2
            const double* Hp = h1 + 7 * nt;
            double q = 0.0;
            for (int a_ = 0; a_ < nt; ++a_) {
                double row = 0.0;
                for (int b = 0; b < nt; ++b) row += Hp[a_ * nt + b] * (h1[b] - h0[b]);
                q += (h1[a_] - h0[a_]) * row;
            }
            if (sqrt(-q > 0.0 ? -q : 0.0) < o->theta_rtol) break;
        }
        const int z0_mode = (i > 1 || o->z0_warm) ? MUSE_Z0_WARM : MUSE_Z0_ZERO;
        // every iteration re-draws the same streams at a new theta (src/muse.jl:134,169): the first one stores the
        // standard normals, the later ones load them instead of running the generator again
        rc = map_async_impl(c, seed, 0, S, 1, theta, o->atol, z0_mode, 0, nullptr, i == 1 ? 1 : 2);
        if (rc) return rc;
        rc = muse_batch_wait(c, 0, g.data(), info.data());
        if (rc) return rc;
        double* h = hist_out + (int64_t)(i - 1) * H;
        double* gs = gsims_out + (int64_t)(i - 1) * S * nt;
        memcpy(gs, g.data() + nt, (size_t)S * nt * sizeof(double));
        if (info_out) memcpy(info_out + (int64_t)(i - 1) * (S + 1), info.data(), ((size_t)S + 1) * sizeof(muse_info));
        double Hlike[kMaxTheta * kMaxTheta], Hinv_like_inv[kMaxTheta * kMaxTheta], Hpost[kMaxTheta * kMaxTheta];
        for (int k = 0; k < nt; ++k) {
            double m = 0.0;
            for (int s = 0; s < S; ++s) m += gs[(int64_t)s * nt + k];
            m /= S;
            double v = 0.0;
            for (int s = 0; s < S; ++s) {
                const double dlt = gs[(int64_t)s * nt + k] - m;
                v += dlt * dlt;
            }
            v /= (S - 1);  // corrected (src/muse.jl:188)
            if (o->prior_kind == 1) {
                const double sg2 = o->prior_sigma[k] * o->prior_sigma[k];
                gprior[k] = -(theta[k] - o->prior_mean[k]) / sg2;
                hprior[k] = -1.0 / sg2;
            } else {
                gprior[k] = 0.0;
                hprior[k] = 0.0;
            }
            h[k] = theta[k];
            h[nt + k] = g[k];                    // g_like_dat
            h[2 * nt + k] = g[k] - m;            // g_like  = g_dat - mean(g_sims)
            h[3 * nt + k] = gprior[k];
            h[4 * nt + k] = h[2 * nt + k] + gprior[k];  // g_post
            h[5 * nt + k] = -1.0 / v;            // diag H^-1_like
            h[6 * nt + k] = hprior[k];
        }
        // H^-1_post = inv(inv(H^-1_like) + H_prior): both diagonal here, kept general through the dense inverse
        for (int a_ = 0; a_ < nt * nt; ++a_) Hlike[a_] = 0.0;
        for (int k = 0; k < nt; ++k) Hlike[k * nt + k] = h[5 * nt + k];
        if (!small_inverse(nt, Hlike, Hinv_like_inv)) return fail(MUSE_ERR_INVALID, "muse_run: singular H^-1_like (zero score variance)");
        for (int k = 0; k < nt; ++k) Hinv_like_inv[k * nt + k] += hprior[k];
        if (!small_inverse(nt, Hinv_like_inv, Hpost)) return fail(MUSE_ERR_INVALID, "muse_run: singular posterior Hessian");
        for (int a_ = 0; a_ < nt * nt; ++a_) h[7 * nt + a_] = Hpost[a_];
        for (int a_ = 0; a_ < nt; ++a_) {  // Newton-Raphson step (src/muse.jl:224)
            double stp = 0.0;
            for (int b = 0; b < nt; ++b) stp += Hpost[a_ * nt + b] * h[4 * nt + b];
            theta[a_] = h[a_] - o->alpha * stp;
        }
        const double t_end = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                 std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-9;
        h[7 * nt + nt * nt] = t_end - t_start;
        n = i;
    }
    *niter_out = n;
    for (int k = 0; k < nt; ++k) theta_out[k] = theta[k];
    return MUSE_OK;
}

int muse_get_zhat(muse_ctx* c, int64_t b, int64_t e, double* out, int mem) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!out || b < 0 || e < b || e > c->zhat_slots) return fail(MUSE_ERR_INVALID, "bad slot range");
    if (e == b) return MUSE_OK;
    HIPCHK(hipMemcpy2DAsync(out, (size_t)c->N * sizeof(double), c->zhat + b * c->ld, (size_t)c->ld * sizeof(double),
                            (size_t)c->N * sizeof(double), (size_t)(e - b), out_kind(mem), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MUSE_OK;
}
int muse_set_zhat(muse_ctx* c, int64_t b, int64_t e, const double* in, int mem) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!in || b < 0 || e < b) return fail(MUSE_ERR_INVALID, "bad slot range");
    if (e == b) return MUSE_OK;
    rc = ensure_zhat(c, e);
    if (rc) return rc;
    HIPCHK(hipMemcpy2DAsync(c->zhat + b * c->ld, (size_t)c->ld * sizeof(double), in, (size_t)c->N * sizeof(double),
                            (size_t)c->N * sizeof(double), (size_t)(e - b), in_kind(mem), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MUSE_OK;
}

int muse_fd_jacobian_batch(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, const double* theta0,
                           const double* step, double atol, int fid_mode, int64_t fid_sim, double* Hs_out,
                           muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !step || !Hs_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (sim_end < sim_begin || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad sim range");
    if (fid_mode != 0 && fid_mode != 1) return fail(MUSE_ERR_INVALID, "fid_mode must be 0 or 1");
    const int64_t nsims = sim_end - sim_begin;
    if (nsims == 0) return MUSE_OK;
    const int nt = c->ntheta;
    const int64_t n = nsims * 2 * nt;
    if (n > 0x7fffffff) return fail(MUSE_ERR_INVALID, "batch too large");
    for (int j = 0; j < nt; ++j)
        if (!(step[j] != 0.0) || !isfinite(step[j])) return fail(MUSE_ERR_INVALID, "step must be finite and non-zero");
    // 1. fiducial MAPs at theta0 from zero(z) (src/muse.jl:417-423)
    const int64_t nfid = fid_mode == 0 ? 1 : nsims;
    rc = ensure_zhat(c, nfid);
    if (rc) return rc;
    // every simulation is drawn 2*ntheta times (same randoms, perturbed theta; src/muse.jl:426-432): its standard
    // normals are generated once -- by its own fiducial problem (fid_mode 1) or by a normals-only element of the
    // fiducial launch (fid_mode 0) -- and loaded by the perturbed problems
    const bool cached = ensure_ncache(c, nsims);
    const int64_t nprep = nfid + ((cached && fid_mode == 0) ? nsims : 0);
    rc = ensure_results(c, 1, n > nprep ? n : nprep);
    if (rc) return rc;
    {
        BatchArgs a;
        base_args(c, a, theta0);
        a.kind = BATCH_STD;
        a.seed = seed;
        a.atol = atol;
        a.nproblems = (int)nprep;
        a.nstd = (int)nfid;
        a.norm_sim0 = sim_begin;
        if (cached) {
            a.ncache = c->ncache;
            a.ncache_sim0 = sim_begin;
            a.ncache_count = (int)nsims;
            a.ncache_mode = 1;
        }
        a.include_data = 0;
        a.z0_mode = MUSE_Z0_ZERO;
        a.store_zhat = 1;
        a.sim_begin = fid_mode == 0 ? fid_sim : sim_begin;
        a.slot0 = 0;
        a.scores = c->scores_dev[1];
        a.info = c->info_dev[1];
        rc = launch_batch(c, a);
        if (rc) return rc;
    }
    // 2. the 2*ntheta perturbed simulations per sim, MAP and score at theta0
    std::vector<double> th(nt);
    for (int j = 0; j < nt; ++j) {
        for (int s = 0; s < 2; ++s) {
            for (int k = 0; k < nt; ++k) th[k] = theta0[k];
            th[j] = theta0[j] + (s == 0 ? step[j] : -step[j]);
            make_thetaset(c, th.data(), c->tsample_pin[2 * j + s]);
        }
    }
    HIPCHK(hipMemcpyAsync(c->tsample_dev, c->tsample_pin, (size_t)2 * nt * sizeof(ThetaSet), hipMemcpyHostToDevice,
                          c->stream));
    {
        BatchArgs a;
        base_args(c, a, theta0);
        a.kind = BATCH_FD;
        a.seed = seed;
        a.atol = atol;
        a.nproblems = (int)n;
        a.sim_begin = sim_begin;
        a.fid_slot = fid_mode == 0 ? 0 : -1;
        a.slot0 = 0;
        a.tsample = c->tsample_dev;
        a.scores = c->scores_dev[1];
        a.info = c->info_dev[1];
        if (cached) {
            a.ncache = c->ncache;
            a.ncache_sim0 = sim_begin;
            a.ncache_count = (int)nsims;
            a.ncache_mode = 2;
        }
        rc = launch_batch(c, a);
        if (rc) return rc;
    }
    rc = enqueue_results_copy(c, 1, n);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    const double* g = c->scores_pin[1];
    for (int64_t s = 0; s < nsims; ++s)
        for (int j = 0; j < nt; ++j) {
            const double* gp = g + ((s * nt + j) * 2 + 0) * nt;
            const double* gm = g + ((s * nt + j) * 2 + 1) * nt;
            for (int i = 0; i < nt; ++i) Hs_out[(s * nt + i) * nt + j] = (-0.5 * gm[i] + 0.5 * gp[i]) / step[j];
        }
    if (info_out) memcpy(info_out, c->info_pin[1], (size_t)n * sizeof(muse_info));
    return MUSE_OK;
}

int muse_implicit_H_batch(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, const double* theta0, double atol,
                          int cg_maxiter, double* Hs_out, int32_t* cg_iters_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !Hs_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (sim_end < sim_begin || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad sim range");
    if (cg_maxiter < 1) return fail(MUSE_ERR_INVALID, "cg_maxiter must be >= 1");
    const int64_t nsims = sim_end - sim_begin;
    if (nsims == 0) return MUSE_OK;
    const int nt = c->ntheta;
    if (nsims * nt > 0x7fffffff) return fail(MUSE_ERR_INVALID, "batch too large");
    rc = ensure_zhat(c, 1);
    if (rc) return rc;
    rc = ensure_results(c, 2, nsims * nt);
    if (rc) return rc;
    BatchArgs a;
    base_args(c, a, theta0);
    a.kind = BATCH_IMPLICIT;
    a.seed = seed;
    a.atol = atol;
    a.cg_maxiter = cg_maxiter;
    a.nproblems = (int)nsims;
    a.sim_begin = sim_begin;
    a.slot0 = 0;
    a.scores = c->scores_dev[2];
    a.info = c->info_dev[2];
    rc = launch_batch(c, a);
    if (rc) return rc;
    rc = enqueue_results_copy(c, 2, nsims * nt);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (*c->error_flag) {
        *c->error_flag = 0;
        return fail(MUSE_ERR_HIP, "a cluster wait expired inside the solver kernel");
    }
    memcpy(Hs_out, c->scores_pin[2], (size_t)nsims * nt * nt * sizeof(double));
    if (cg_iters_out)
        for (int64_t k = 0; k < nsims * nt; ++k) cg_iters_out[k] = c->info_pin[2][k].iterations;
    return MUSE_OK;
}

}  // extern "C"
