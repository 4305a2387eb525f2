// vec.hpp -- vector accessors of the two storage policies and the per-thread element loop (see muse_kernels.hip).
#pragma once
#include <type_traits>
#include "args.hpp"

namespace muse {

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
// Cache policy of a buffer access (the builtins' last argument): kPlain, or kCoherent = sc0 | sc1 -- a store that is
// written through to the memory side (and leaves no line behind in the XCD's L2), a load that is not served from this
// CU's L1.  Vectors whose elements OTHER workgroups read inside a launch (the stencil model's x, z, s in cluster
// placements) are stored coherently and their foreign elements loaded coherently: together with the storing waves'
// `s_waitcnt vmcnt(0)` before the cluster's tagged-granule exchange this makes a pass's stores visible to the other
// members without any fence (MI355X_MICROARCH.md, "Valid forms": sc1 stores and loads on both sides).
constexpr int kPlain = 0, kCoherent = 17;
template <int AUX = kPlain>
__device__ __forceinline__ double load_f64(const rsrc_t& rs, int i) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off8(i), 0, AUX);
    return __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
}
template <int AUX = kPlain>
__device__ __forceinline__ void load_f64x2(const rsrc_t& rs, int i, double& d0, double& d1) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off8(i), 0, AUX);
    d0 = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
    d1 = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
}
template <int AUX = kPlain>
__device__ __forceinline__ void store_f64x2(const rsrc_t& rs, int i, double d0, double d1) {
    const long long b0 = __double_as_longlong(d0), b1 = __double_as_longlong(d1);
    u32x4 v;
    v.x = (unsigned)(b0 & 0xffffffffll);
    v.y = (unsigned)(b0 >> 32);
    v.z = (unsigned)(b1 & 0xffffffffll);
    v.w = (unsigned)(b1 >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off8(i), 0, AUX);
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
template <int U, bool COH = false>
struct BufChunk {
    static constexpr int kAux = COH ? kCoherent : kPlain;  // stores and foreign-element loads (get1); own pairs load plain
    rsrc_t rsrc;
    mutable double c0[U], c1[U];
    mutable bool pl = false;  // the trip's pairs are in c0/c1 already (preload): get() issues nothing
    double st[2 * U];
    __device__ __forceinline__ void bind(const double* base, int64_t ld) { rsrc = make_rsrc(base, ld * 8); }
    __device__ __forceinline__ double get(int jj, int i) const {
        if (pl) return (jj & 1) ? c1[jj >> 1] : c0[jj >> 1];
        if ((jj & 1) == 0) {
            double d0;
            load_f64x2(rsrc, i, d0, c1[jj >> 1]);
            return d0;
        }
        return c1[jj >> 1];
    }
    // all of a trip's loads of this vector, issued at once and ahead of anything the caller puts between them and the
    // element bodies (the background generator's arithmetic)
    template <int UU>
    __device__ __forceinline__ void preload(int i0, int pstride) const {
        static_assert(UU <= U, "trip longer than the chunk");
#pragma unroll
        for (int u = 0; u < UU; ++u) load_f64x2(rsrc, i0 + 2 * u * pstride, c0[u], c1[u]);
        pl = true;
    }
    __device__ __forceinline__ void unload() const { pl = false; }
    __device__ __forceinline__ double get1(int i) const { return load_f64<kAux>(rsrc, i); }
    __device__ __forceinline__ void own_pair(int i0, double& a, double& b) const { load_f64x2(rsrc, i0, a, b); }
    __device__ __forceinline__ void set(int jj, int, double d) { st[jj] = d; }
    __device__ __forceinline__ void clear() {}
    // the UU (<= U) pairs i0, i0 + 2*pstride, ... of the trip that started at element i0
    template <int UU>
    __device__ __forceinline__ void flush(int i0, int pstride) {
        static_assert(UU <= U, "trip longer than the staging area");
#pragma unroll
        for (int u = 0; u < UU; ++u) store_f64x2<kAux>(rsrc, i0 + 2 * u * pstride, st[2 * u], st[2 * u + 1]);
    }
};
// The search direction (and the two-loop recursion's q) of the stencil model in a cluster: the workgroup's own elements
// live in LDS -- pair k of thread t at local slot k*T + t -- and are mirrored to the HBM vector only where somebody else
// reads them: the pairs of the two edge lanes of every wave (the neighbouring wave's or workgroup's stencil) and the
// pairs next to the periodic wrap (the patch path reads those element-wise).  The two-loop recursion, which reads and
// writes this vector once per history pair, then touches HBM for the history vectors only.
template <int U, int T>
struct LdsMirror {
    static constexpr int kAux = kCoherent;
    rsrc_t rsrc;  // the HBM mirror
    lds_double* p;
    int tfirst, sh, tid, n;  // pstride = 1 << sh; n = N
    mutable double c1[U];
    double st[2 * U];
    __device__ __forceinline__ void bind(const double* hbm, int64_t ld, double* lds, int tfirst_, int pstride, int tid_, int n_) {
        rsrc = make_rsrc(hbm, ld * 8);
        p = (lds_double*)lds;
        tfirst = tfirst_;
        sh = 31 - __builtin_clz((unsigned)pstride);
        tid = tid_;
        n = n_;
    }
    __device__ __forceinline__ int slot(int i) const { return (((((i >> 1) - tfirst) >> sh) * T) + tid) << 1; }
    __device__ __forceinline__ double get(int jj, int i) const {
        if ((jj & 1) == 0) {
            const lds_double* q = p + slot(i);
            const double d0 = q[0];
            c1[jj >> 1] = q[1];
            return d0;
        }
        return c1[jj >> 1];
    }
    __device__ __forceinline__ void own_pair(int i0, double& a, double& b) const {
        const lds_double* q = p + slot(i0);
        a = q[0];
        b = q[1];
    }
    __device__ __forceinline__ double get1(int i) const { return load_f64<kCoherent>(rsrc, i); }  // a mirrored element
    __device__ __forceinline__ void set(int jj, int, double d) { st[jj] = d; }
    __device__ __forceinline__ void clear() {}
    template <int UU>
    __device__ __forceinline__ void flush(int i0, int pstride) {
        static_assert(UU <= U, "trip longer than the staging area");
        const int lane = tid & 63;
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const int i = i0 + 2 * u * pstride;
            lds_double* q = p + slot(i);
            q[0] = st[2 * u];
            q[1] = st[2 * u + 1];
            if (lane == 0 || lane == 63 || i < 4 || i + 6 >= n) store_f64x2<kCoherent>(rsrc, i, st[2 * u], st[2 * u + 1]);
        }
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
// read-list entry of a pass (Solver::pass_elems_r): a vector the pass reads, under a (workgroup-uniform) condition
template <class V>
struct ReadIf {
    const V& v;
    bool on;
    template <int UU>
    __device__ __forceinline__ void preload(int i0, int pstride) const {
        if (on) v.template preload<UU>(i0, pstride);
    }
    __device__ __forceinline__ void unload() const { v.unload(); }
};
template <class V>
__device__ __forceinline__ ReadIf<V> reads(const V& v, bool on = true) { return ReadIf<V>{v, on}; }
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
        // pstride == T for a single workgroup (the caller passes the constant), csize*T in cluster mode
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int i0 = 2 * (t + j * pstride);
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

// The same loop with a prologue per trip: pre(integral_constant<int, UU>, i0, pstride) runs before the element bodies of
// the UU pairs i0, i0 + 2*pstride, ... (the sampler draws the normals of all of them side by side, rng.hpp); the bodies
// find pair u of the trip at jj % (2U) = 2u, 2u+1.
template <int T, int EPT, int U, class P, class F, class... W>
__device__ __forceinline__ void for_elems_pre(int64_t ld, int tfirst, int pstride, P&& pre, F&& f, W&&... written) {
    int t = tfirst;
    asm volatile("" : "+v"(t));
    if constexpr (EPT > 0) {
#pragma unroll
        for (int j0 = 0; j0 < EPT; j0 += U) {
            const int i0 = 2 * (t + j0 * pstride);
            if (j0 + U <= EPT) {
                pre(std::integral_constant<int, U>{}, i0, pstride);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    f(2 * (j0 + u), i0 + 2 * u * pstride);
                    f(2 * (j0 + u) + 1, i0 + 2 * u * pstride + 1);
                }
            } else {
                pre(std::integral_constant<int, (EPT % U) ? (EPT % U) : U>{}, i0, pstride);
#pragma unroll
                for (int u = 0; u < EPT % U; ++u) {
                    f(2 * (j0 + u), i0 + 2 * u * pstride);
                    f(2 * (j0 + u) + 1, i0 + 2 * u * pstride + 1);
                }
            }
        }
    } else {
        const int n = (int)ld;
#pragma unroll 1
        for (int i0 = 2 * t; i0 < n; i0 += 2 * U * pstride) {
            pre(std::integral_constant<int, U>{}, i0, pstride);
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

}  // namespace muse
