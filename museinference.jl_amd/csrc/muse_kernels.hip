// muse_kernels.hip -- MI355X (gfx950) engine for the MUSE inner loop: one persistent workgroup per
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
//
// This translation unit holds ALL device code (the solver kernel in every model x placement instantiation, the
// per-simulation operator kernels) and the shims that launch it; the host side -- context, workspace, C ABI -- is
// muse_engine.cpp.
#include <hip/hip_runtime.h>
#include <math.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "../../include/muse_hip.h"

#include "kernels.hpp"

namespace muse {

// ------------------------------------------------------------------------------------------------
// Per-simulation operator kernels (API parity with the reference's per-sim interface; not the
// performance path).
// The per-block tables of a launch whatever its tier (args.hpp, BigTheta): BatchArgs is the first kernel parameter.
__device__ __forceinline__ const BigTheta* kernarg_big() {
    return reinterpret_cast<const BigTheta*>(reinterpret_cast<const char*>((const void*)__builtin_amdgcn_kernarg_segment_ptr()) +
                                             offsetof(BatchArgs, big));
}
__device__ __forceinline__ int any_block(const BatchArgs& a, int64_t i) {
    if (a.ntheta > kMaxTheta) return block_of_big((int)a.N, a.ntheta, 1.0 / (double)a.N, (int)i);
    return a.ntheta > 1 ? block_of(a, i) : 0;
}
__device__ __forceinline__ double any_sd(const BatchArgs& a, int k) { return a.ntheta > kMaxTheta ? kernarg_big()->sd[k] : a.cur.t.sd[k]; }
__device__ __forceinline__ double any_iv(const BatchArgs& a, int k) { return a.ntheta > kMaxTheta ? kernarg_big()->iv[k] : a.cur.t.iv[k]; }

template <int MODEL>
__global__ void __launch_bounds__(256) sample_kernel(BatchArgs a, uint64_t sim, double* __restrict__ x,
                                                     double* __restrict__ z) {
    const int64_t N = a.N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = any_block(a, i);
        const double sdk = any_sd(a, k);
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
#ifdef MUSE_USER_MODEL_HEADER
__global__ void __launch_bounds__(256) sample_user_kernel(BatchArgs a, uint64_t sim, double* __restrict__ x, double* __restrict__ z) {
    const int64_t N = a.N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = any_block(a, i);
        const NormalPair np = normal_pair(a.seed, sim, (uint64_t)i);
#ifdef MUSE_MODEL_PAIR
        PairS c;   // (models.hpp, pair_table: a block's coefficients side by side)
        c.c[0] = pair_table(a.cur.t)[4 * k];
        c.c[1] = pair_table(a.cur.t)[4 * k + 1];
        UserModel<kMaxTheta>::sample(c, np.n1, np.n2, z[i], x[i], (int)i);
#else
        UserModel<1>::sample(any_sd(a, k), np.n1, np.n2, z[i], x[i], (int)i);
#endif
    }
}
#endif
// The standard normals of one stream, one element per thread (rng.hpp: the same function of (seed, sim, element) the solver's
// generator evaluates, so a problem that LOADS these draws the same bits it would have generated).
__global__ void __launch_bounds__(256) normals_kernel(uint64_t seed, uint64_t sim, int64_t ld, double* __restrict__ n1, double* __restrict__ n2) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ld; i += (int64_t)gridDim.x * blockDim.x) {
        const NormalPair np = normal_pair(seed, sim, (uint64_t)i);
        n1[i] = np.n1;
        n2[i] = np.n2;
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
    constexpr bool kBig = Model::MAXB > kMaxTheta;       // (the big tier: block sums eight at a time, as Solver::finish)
    constexpr int T = 1024, MAXB = kBig ? 8 : Model::MAXB;
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
        const int k = MAXB > 1 ? any_block(a, i) : 0;
        if constexpr (Model::kPair) {   // two parameters per block (include/muse_model.h): four coefficients, two block sums
            const int K = a.ntheta >> 1;
            PairGp c;   // (the block's record and the element's validity: models.hpp -- the pad element contributes nothing)
            c.p = pair_table(a.cur.t) + 4 * k;
            c.valid = i < Ni;
            const double gp = Model::grad(c, xin[i], zin[i], sum[0], i);
            if (gout) gout[i] = -gp;
            double t0, t1;
            Model::score_terms(c, xin[i], zin[i], t0, t1, i);
#pragma unroll
            for (int b = 0; b < MAXB; ++b) acc[b] += (k == b) ? t0 : ((k + K == b) ? t1 : 0.0);
            return;
        } else {
        const double ivk = any_iv(a, k);
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
            gi = Model::grad(ivk, xin[i], zin[i], sum[0], i);
        }
        if (gout) gout[i] = -gi;
        const double t = Model::score_term(xin[i], zin[i], i);
#pragma unroll
        for (int b = 0; b < MAXB; ++b) acc[b] += (k == b) ? t : 0.0;
        }
    });
    block_allreduce<T, 2, 0>(sum, mx, red, parity, tid);
    block_allreduce<T, MAXB, 0>(acc, mx, red, parity, tid);
    auto count = [&](int k) {  // elements of block k: bnd[k] = ceil(k N / nblocks)
        const int64_t nt = Model::kPair ? a.ntheta >> 1 : a.ntheta;
        return (double)(((int64_t)(k + 1) * N + nt - 1) / nt - ((int64_t)k * N + nt - 1) / nt);
    };
    if (tid == 0) {
        out[0] = -(0.5 * (sum[0] + a.cur.f_const));
        if constexpr (Model::kPair) {
            const int K = a.ntheta >> 1;
            for (int k = 0; k < K; ++k) Model::score(pair_table(a.cur.t) + 4 * k, acc[k], acc[K + k], count(k), out[1 + k], out[1 + K + k]);
        } else {
        for (int b = 0; b < MAXB; ++b)
            if (b < a.ntheta) out[1 + b] = 0.5 * (any_iv(a, b) * acc[b] - count(b));
        }
    }
    if constexpr (kBig) {
        for (int c = 8; c < a.ntheta; c += 8) {
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[b] = 0.0;
            for_elems<T, 0, 1>(a.ld, tid, T, [&](int, int i) {
                const int k = any_block(a, i) - c;
                const double t = Model::score_term(xin[i], zin[i], i);
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[b] += (k == b) ? t : 0.0;
            });
            block_allreduce<T, 8, 0>(acc, mx, red, parity, tid);
            if (tid == 0)
                for (int b = 0; b < 8; ++b)
                    if (c + b < a.ntheta) out[1 + c + b] = 0.5 * (any_iv(a, c + b) * acc[b] - count(c + b));
        }
    }
}


}  // namespace muse

// ================================================================================================
// Dispatch.  The kernels themselves are instantiated in the units of kernels_part.hip (kernels.hpp: MUSE_PART_n); a
// single-instantiation development build (MUSE_INSPECT, tools/regs.py) instantiates its one kernel right here.
// ================================================================================================
namespace muse {

#if !defined(MUSE_INSPECT) && !defined(MUSE_INSPECT_LOOP)
MUSE_PART_0(extern) MUSE_PART_1(extern) MUSE_PART_2(extern) MUSE_PART_3(extern) MUSE_PART_4(extern) MUSE_PART_5(extern) MUSE_PART_6(extern) MUSE_PART_7(extern)
#endif

hipError_t launch_solver(const LaunchShape& s, const BatchArgs& a, hipStream_t st) {
#ifdef MUSE_INSPECT  // development aid (tools/regs.py --check): instantiate ONE kernel, for a quick look at its assembly
    return launch_one<MUSE_INSPECT>(s, a, st);
#elif defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)   // ... of the two-parameter family: 2 ... kMaxTheta components, no big tier,
    if (s.model != MUSE_MODEL_USER || s.big || s.implicit) return hipErrorInvalidValue;   // no implicit differentiation
    return s.ntheta == 2 ? launch_place<UserModel<2>>(s, a, st) : s.ntheta <= 4 ? launch_place<UserModel<4>>(s, a, st)
                                                                                : launch_place<UserModel<kMaxTheta>>(s, a, st);
#elif defined(MUSE_USER_MODEL_HEADER)  // a library built from a user's model header holds that model only (user_model.hpp)
    if (s.model != MUSE_MODEL_USER) return hipErrorInvalidValue;
    if (s.big && !s.implicit) return launch_place_big<UserModel<kBigTheta>>(s, a, st);
    if (s.implicit) {
#ifdef MUSE_MODEL_SECOND
        if (s.big) return launch_place_implicit<UserModel<kBigTheta>>(s, a, st);
        return s.ntheta == 1 ? launch_place_implicit<UserModel<1>>(s, a, st) : launch_place_implicit<UserModel<kMaxTheta>>(s, a, st);
#else
        return hipErrorInvalidValue;  // (muse_engine.cpp refuses the call before it gets here)
#endif
    }
    return s.ntheta == 1 ? launch_place<UserModel<1>>(s, a, st) : launch_place<UserModel<kMaxTheta>>(s, a, st);
#else
    const int nt = s.ntheta;
    if (s.big) {  // the big tier (args.hpp, BigTheta; muse_engine.cpp, tier_big): streaming placements only
        if (s.model == MUSE_MODEL_FUNNEL)
            return s.implicit ? launch_place_implicit<FunnelModel<kBigTheta>>(s, a, st) : launch_place_big<FunnelModel<kBigTheta>>(s, a, st);
        if (s.model == MUSE_MODEL_SMOOTH)
            return s.implicit ? launch_place_implicit<SmoothModel<kBigTheta>>(s, a, st) : launch_place_big<SmoothModel<kBigTheta>>(s, a, st);
        return hipErrorInvalidValue;
    }
    if (s.implicit) {
        if (s.model == MUSE_MODEL_NOISE) return launch_place_implicit<NoiseModel>(s, a, st);
        if (s.model == MUSE_MODEL_FUNNEL)
            return nt == 1   ? launch_place_implicit<FunnelModel<1>>(s, a, st)
                   : nt == 2 ? launch_place_implicit<FunnelModel<2>>(s, a, st)
                   : nt <= 4 ? launch_place_implicit<FunnelModel<4>>(s, a, st)
                             : launch_place_implicit<FunnelModel<kMaxTheta>>(s, a, st);
        return nt <= 2   ? launch_place_implicit<SmoothModel<2>>(s, a, st)
               : nt <= 4 ? launch_place_implicit<SmoothModel<4>>(s, a, st)
                         : launch_place_implicit<SmoothModel<kMaxTheta>>(s, a, st);
    }
    if (s.model == MUSE_MODEL_NOISE) return launch_place<NoiseModel>(s, a, st);
    if (s.model == MUSE_MODEL_FUNNEL)
        return nt == 1   ? launch_place<FunnelModel<1>>(s, a, st)
               : nt == 2 ? launch_place<FunnelModel<2>>(s, a, st)
               : nt <= 4 ? launch_place<FunnelModel<4>>(s, a, st)
                         : launch_place<FunnelModel<kMaxTheta>>(s, a, st);
    return nt <= 2   ? launch_place<SmoothModel<2>>(s, a, st)
           : nt <= 4 ? launch_place<SmoothModel<4>>(s, a, st)
                     : launch_place<SmoothModel<kMaxTheta>>(s, a, st);
#endif
}

static hipError_t loop_dispatch(const LaunchShape& s, const LoopCall& c) {
#ifdef MUSE_INSPECT_LOOP  // development aid (tools/regs.py --check): ONE loop kernel
    return loop_one<MUSE_INSPECT_LOOP>(s, c);
#elif defined(MUSE_INSPECT)
    return hipErrorNotSupported;
#elif defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
    if (s.model != MUSE_MODEL_USER) return hipErrorInvalidValue;
    return s.ntheta == 2 ? loop_place<UserModel<2>>(s, c) : s.ntheta <= 4 ? loop_place<UserModel<4>>(s, c) : loop_place<UserModel<kMaxTheta>>(s, c);
#elif defined(MUSE_USER_MODEL_HEADER)
    if (s.model != MUSE_MODEL_USER) return hipErrorInvalidValue;
    return s.ntheta == 1 ? loop_place<UserModel<1>>(s, c) : loop_place<UserModel<kMaxTheta>>(s, c);
#else
    const int nt = s.ntheta;
    if (s.model == MUSE_MODEL_NOISE) return loop_place<NoiseModel>(s, c);
    if (s.model == MUSE_MODEL_FUNNEL)
        return nt == 1   ? loop_place<FunnelModel<1>>(s, c)
               : nt == 2 ? loop_place<FunnelModel<2>>(s, c)
               : nt <= 4 ? loop_place<FunnelModel<4>>(s, c)
                         : loop_place<FunnelModel<kMaxTheta>>(s, c);
    return hipErrorNotSupported;  // the stencil model streams
#endif
}
bool loop_supported(const LaunchShape& s) {
    LoopCall c{LOOP_QUERY, nullptr, nullptr, nullptr, 0, nullptr};
    return loop_dispatch(s, c) == hipSuccess;
}
hipError_t loop_max_grid(const LaunchShape& s, int num_cus, int* max_grid, int* scratch_bytes) {
    LoopCall c{LOOP_GRID, nullptr, nullptr, nullptr, num_cus, max_grid, scratch_bytes};
    return loop_dispatch(s, c);
}
hipError_t launch_loop(const LaunchShape& s, const BatchArgs& a, const LoopArgs& l, hipStream_t st) {
    LoopCall c{LOOP_LAUNCH, &a, &l, st, 0, nullptr, nullptr};
    return loop_dispatch(s, c);
}
// LDS of the loop kernel beside the map kernel's: the persistent block, and -- where x and g are not in LDS -- the step's arrays
size_t loop_extra_lds(bool xg_lds, int64_t nprob, int ntheta) {
    const size_t step = ((size_t)nprob * ntheta + 24) * sizeof(double) + sizeof(StepWork);
    return (size_t)kLoopPersistDoubles * sizeof(double) + (xg_lds ? 0 : step);
}
size_t loop_step_bytes(int64_t nprob, int ntheta) { return ((size_t)nprob * ntheta + 24) * sizeof(double) + sizeof(StepWork); }

// ---- the score boards' set-up hand-shake (muse_comm.cpp: setup_boards; include/muse_hip.h: muse_comm_board_status) ---------------
// ONE wavefront per rank.  Lane q < nstore stores this rank's tagged granule pair into board q -- the system-scope relaxed 8-byte
// stores of Solver::finish (solver.hpp: gran_sys; the boards in device memory: one per rank, nstore = nranks; the board in pinned
// host memory: the one, nstore = 1) -- and lane q < nranks then polls ITS OWN GPU's view of the board for rank q's pair with the
// stepper's 16-byte coherent buffer load (kernels.hpp), until the pair carries the tag and this hand-shake's payload or `ticks` of
// the 100 MHz counter have passed.  result (pinned host memory): [0..1] bit mask of the ranks seen, [2] ticks until the last one
// landed (the bound when it expired).  The slots are a region of their own behind the score granules.
struct HandshakeArgs {
    unsigned long long* own;
    unsigned long long* store[8];
    int nstore, nranks, rank;
    unsigned int tag;
    unsigned long long slot0;      // granule index of rank 0's pair (rank r: slot0 + 2 r)
    unsigned long long ticks;
    unsigned int* result;
};
__global__ void __launch_bounds__(64) board_handshake_kernel(HandshakeArgs h) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) unsigned int gu32;
    const int lane = (int)threadIdx.x;
    const unsigned long long tg = (unsigned long long)h.tag << 32;
    if (lane < h.nstore) {
        gu64* pq = (gu64*)h.store[lane] + h.slot0 + 2 * (unsigned long long)h.rank;
        __hip_atomic_store(pq, tg | (unsigned long long)(unsigned)(h.rank + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(pq + 1, tg | (unsigned long long)(unsigned)(0x5a5a0000 + h.rank), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const rsrc_t grs = make_rsrc(h.own, (int64_t)(h.slot0 + 128) * 8);
    bool seen = lane >= h.nranks;
    unsigned long long t0, now;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    now = t0;
    for (;;) {
        if (!seen) {
            double lo, hi;
            load_f64x2<kCoherent>(grs, (int)h.slot0 + 2 * lane, lo, hi);
            const unsigned long long glo = (unsigned long long)__double_as_longlong(lo), ghi = (unsigned long long)__double_as_longlong(hi);
            seen = glo == (tg | (unsigned long long)(unsigned)(lane + 1)) && ghi == (tg | (unsigned long long)(unsigned)(0x5a5a0000 + lane));
        }
        const bool all = __builtin_amdgcn_ballot_w64(!seen) == 0ull;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        if (all || now - t0 > h.ticks) break;
        __builtin_amdgcn_s_sleep(4);
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(seen && lane < h.nranks);
    if (lane == 0) {
        __hip_atomic_store((gu32*)h.result, (unsigned)(mask & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((gu32*)h.result + 1, (unsigned)(mask >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((gu32*)h.result + 2, (unsigned)(now - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
hipError_t launch_board_handshake(unsigned long long* own, unsigned long long* const* store, int nstore, int nranks, int rank, unsigned int tag,
                                  unsigned long long slot0, unsigned long long ticks, unsigned int* result, hipStream_t st) {
    if (nstore < 1 || nstore > 8 || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return hipErrorInvalidValue;
    HandshakeArgs h;
    memset(&h, 0, sizeof h);
    h.own = own;
    for (int q = 0; q < nstore; ++q) h.store[q] = store[q];
    h.nstore = nstore;
    h.nranks = nranks;
    h.rank = rank;
    h.tag = tag;
    h.slot0 = slot0;
    h.ticks = ticks;
    h.result = result;
    hipLaunchKernelGGL(board_handshake_kernel, dim3(1), dim3(64), 0, st, h);
    return hipGetLastError();
}

hipError_t launch_normals(uint64_t seed, uint64_t sim, int64_t ld, double* slot, hipStream_t st) {
    const int grid = (int)((ld + 255) / 256 < 4096 ? (ld + 255) / 256 : 4096);
    hipLaunchKernelGGL(normals_kernel, dim3(grid), dim3(256), 0, st, seed, sim, ld, slot, slot + ld);
    return hipGetLastError();
}

hipError_t launch_sample(int model, const BatchArgs& a, uint64_t sim, double* x, double* z, double* noise, hipStream_t st) {
    const int grid = (int)((a.N + 255) / 256 < 4096 ? (a.N + 255) / 256 : 4096);
#ifdef MUSE_USER_MODEL_HEADER
    (void)noise;
    if (model != MUSE_MODEL_USER) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sample_user_kernel, dim3(grid), dim3(256), 0, st, a, sim, x, z);
    return hipGetLastError();
#endif
    if (model == MUSE_MODEL_NOISE) hipLaunchKernelGGL(sample_kernel<MUSE_MODEL_NOISE>, dim3(grid), dim3(256), 0, st, a, sim, x, z);
    else if (model == MUSE_MODEL_FUNNEL) hipLaunchKernelGGL(sample_kernel<MUSE_MODEL_FUNNEL>, dim3(grid), dim3(256), 0, st, a, sim, x, z);
    else {
        hipLaunchKernelGGL(sample_kernel<MUSE_MODEL_SMOOTH>, dim3(grid), dim3(256), 0, st, a, sim, noise, z);
        hipLaunchKernelGGL(smooth_finish_kernel, dim3(grid), dim3(256), 0, st, a.N, z, noise, x);
    }
    return hipGetLastError();
}

hipError_t launch_loglike(int model, const BatchArgs& a, const double* x, const double* z, double* g, double* out, hipStream_t st) {
#ifdef MUSE_USER_MODEL_HEADER
    if (model != MUSE_MODEL_USER) return hipErrorInvalidValue;
#ifdef MUSE_MODEL_PAIR
    if (a.ntheta > kMaxTheta) return hipErrorInvalidValue;
#else
    if (a.ntheta > kMaxTheta) hipLaunchKernelGGL(loglike_kernel<UserModel<kBigTheta>>, dim3(1), dim3(1024), 0, st, a, x, z, g, out);
    else
#endif
    hipLaunchKernelGGL(loglike_kernel<UserModel<kMaxTheta>>, dim3(1), dim3(1024), 0, st, a, x, z, g, out);
    return hipGetLastError();
#endif
    if (model == MUSE_MODEL_NOISE) hipLaunchKernelGGL(loglike_kernel<NoiseModel>, dim3(1), dim3(1024), 0, st, a, x, z, g, out);
    else if (model == MUSE_MODEL_FUNNEL && a.ntheta > kMaxTheta) hipLaunchKernelGGL(loglike_kernel<FunnelModel<kBigTheta>>, dim3(1), dim3(1024), 0, st, a, x, z, g, out);
    else if (model == MUSE_MODEL_FUNNEL) hipLaunchKernelGGL(loglike_kernel<FunnelModel<kMaxTheta>>, dim3(1), dim3(1024), 0, st, a, x, z, g, out);
    else if (a.ntheta > kMaxTheta) hipLaunchKernelGGL(loglike_kernel<SmoothModel<kBigTheta>>, dim3(1), dim3(1024), 0, st, a, x, z, g, out);
    else hipLaunchKernelGGL(loglike_kernel<SmoothModel<kMaxTheta>>, dim3(1), dim3(1024), 0, st, a, x, z, g, out);
    return hipGetLastError();
}

}  // namespace muse
