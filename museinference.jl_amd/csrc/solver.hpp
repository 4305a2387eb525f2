// solver.hpp -- one element of a batch: sample_x_z -> L-BFGS/HagerZhang MAP over z -> grad_theta score (see muse_kernels.hip).
#pragma once
#include <math.h>

#include "models.hpp"
#include "reduce.hpp"
#include "rng.hpp"

namespace muse {

// ------------------------------------------------------------------------------------------------
// A workgroup's NEXT problem, fetched ahead of the point where its theta is known (the device-resident muse! loop,
// muse_loop_kernel, LDS-resident layout -- whose x and g areas are dead between two iterations): the simulation's cached
// standard normals travel straight into LDS (n2 into the x area, n1 into the g area; for the data element the data vector
// into the x area) by LDS-DMA loads, which occupy no register.  (The warm start is loaded by begin(): prefetched into
// registers it was spilled across the wait for theta -- a load, a wait, a scratch store and a scratch load instead of the
// one load -- and there is no third LDS area.)  Nothing here depends on theta; begin() turns the normals into x in place.  A thread's pair (tid + j T) lands at its own
// slot -- LDS-DMA writes lane l's 16 bytes at the wave's base + 16 l -- so a wave only ever waits for its own loads.
template <int EPT>
struct Prefetch {
    int p;          // the problem the areas' contents belong to, -1: none
    bool have_n1;   // the g area holds the normals n1 of a simulation (X_SAMPLE with a cache slot)
    bool have_n2;   // the x area holds its n2
    bool have_x;    // the x area holds the data vector (X_DATA)
    bool have_xg;   // the g area holds the data vector: sent there while another problem was being solved (begin() moves it over)
    bool g_pending; // n1 was sent to the g area while another problem is being solved: a solve that comes to need g
                    // (a second L-BFGS iteration) waits for the loads to land and drops the prefetch (Solver::drop_g_prefetch)
};
// Cache policy of the cached normals' loads.  They are read once per outer iteration by one compute unit; `nt` (aux = 2) would
// keep them from displacing the MAPs the same compute unit stored for the next iteration's warm starts from its XCD's L2 --
// measured (round 4, one box, interleaved): SLOWER, host loop 53.0 against 46-49 us per iteration, loop kernel 44.6 against
// 42-46.  Plain loads stay; the knob is kept for the next idea.
#ifndef MUSE_NORMALS_AUX
#define MUSE_NORMALS_AUX 0
#endif
constexpr int kNormalsAux = MUSE_NORMALS_AUX;
typedef __attribute__((address_space(1))) const void* glds_src_t;
typedef __attribute__((address_space(3))) void* glds_dst_t;
// both: between two iterations (x and g areas free): n2 -> x area (or the data vector), n1 -> g area.  !both: while another
// problem is being solved, whose x lives in the x area: n1 -> g area only (a solve of one L-BFGS iteration never touches g).
template <int T, int EPT>
__device__ __forceinline__ void prefetch_issue(const BatchArgs& a, int tid, int p, double* lds_x, double* lds_g, Prefetch<EPT>& pf,
                                               bool both) {
    const ProblemDesc d = describe(a, p);
    const int64_t ld = a.ld;
    const bool sim = d.x_mode == X_SAMPLE && d.nslot >= 0;
    pf.p = p;
    pf.have_n1 = sim;
    pf.have_n2 = sim && both;
    pf.have_x = d.x_mode == X_DATA && both;
    pf.have_xg = d.x_mode == X_DATA && !both && !(a.debug & 256);   // (debug bit 8, a tuning aid: the data vector is not sent ahead)
    pf.g_pending = (sim || pf.have_xg) && !both;
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const double* src_x = sim ? a.ncache + (int64_t)(2 * d.nslot + 1) * ld : a.x_data;   // n2, or the data
    const double* src_g = pf.have_xg ? a.x_data : a.ncache + (int64_t)(2 * d.nslot) * ld;   // n1 (or the data on its way through the g area)
    const int wave0 = __builtin_amdgcn_readfirstlane(tl) & ~63;   // the wave's first thread
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i0 = 2 * (tl + j * T);
        if (i0 < (int)ld) {   // (phantom pairs stay out of LDS: they would land beyond the vector)
            const int base = 2 * (wave0 + j * T);         // the wave's first element of this row: wave-uniform
            if (pf.have_n2 || pf.have_x) __builtin_amdgcn_global_load_lds((glds_src_t)(src_x + i0), (glds_dst_t)(lds_x + base), 16, 0, kNormalsAux);
            if (pf.have_n1 || pf.have_xg) __builtin_amdgcn_global_load_lds((glds_src_t)(src_g + i0), (glds_dst_t)(lds_g + base), 16, 0, kNormalsAux);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// (the two-parameter family with ONE block: its four coefficients and the draw's two live in registers like iv0 / sd0 -- members of
//  an empty base for every other model, so that the Solver of the built-in models is laid out as it always was)
template <bool PAIR>
struct PairState {};
template <>
struct PairState<true> {
    double pc[4], psd[2];
};
template <class Model, class Place>
struct Solver : PairState<Model::kPair> {
    static constexpr int T = Place::T, EPT = Place::EPT, U = Place::U, MAXB = Model::MAXB;
    // The big tier (kMaxTheta < ntheta <= kBigTheta, streaming placements only): the per-block coefficients come from the
    // kernel-argument segment (BatchArgs::big) or a finite-difference batch's sampling entry instead of the LDS argument block,
    // the block of an element from arithmetic instead of a compare chain, the block sums eight at a time (block_sums_big).
    static constexpr bool kBig = MAXB > kMaxTheta;
    static constexpr int KB = kBig ? 1 : MAXB;  // per-block accumulators a thread holds
    static_assert(!kBig || !Place::kResident, "the big tier runs in the streaming placements");
    const double* big_sd;  // [ntheta] sd of the theta this problem is DRAWN at
    const double* big_iv;  // [ntheta] exp(-theta) of the MAP's theta
    double big_rcpN;
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
    __device__ __forceinline__ int ps() const { return Place::kCluster ? pstride : T; }  // a constant without clusters
    int crank, csize;          // this workgroup's rank in its cluster, cluster size
    unsigned int cl_epoch;     // cluster reductions done so far in this launch
    bool cl_aborted;           // a bounded wait expired: stop waiting
    double* cl_part;           // this cluster's partial slots [2][csize][8] / granules [2][csize][8][2]
    double* exch;              // LDS [kMaxCluster][8]: the epoch's values of every member (granule exchange)

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
        cl_part = nullptr;
        exch = nullptr;
        gen.p = -1;
        gen.trip = 0;
        gen.active = false;
        bufsel = 0;
        next_p = -1;
    }

    // Cluster all-reduce.  For the elementwise models only these K scalars cross workgroups (every thread reads and
    // writes its own elements of every vector, in every pass); the stencil model's passes also read neighbours that
    // other workgroups own, and those travel by write-through stores and cache-bypassing loads (vec.hpp: kCoherent),
    // drained before the exchange (reduce(), pass_barrier()).  Either way no fence is involved: no L2 write-back,
    // no L1 invalidate.  A value travels as two 8-byte granules {32-bit half, 32-bit tag}, each written by ONE
    // write-through store (MI355X_MICROARCH.md, price list row "handoff-1to1": an aligned 8-byte granule is not
    // torn, and data-tagged granules need no ordering against a separate flag).  The tag is the cluster's epoch
    // number, which keeps growing from launch to launch (BatchArgs::cl_state), so a granule is this epoch's exactly
    // when its tag says so.  Wave 0 publishes the workgroup's 2K granules with one store instruction and then sweeps
    // the cluster's csize*2K granules (one 8-byte L1-bypassing load per lane and sweep) until every tag matches; the
    // other waves get the values through LDS.  Buffers alternate by epoch parity: a workgroup reaches epoch e+2 only
    // after every member has published e+1, i.e. after every member has consumed epoch e.  Combination in rank order.
    template <int KS, int KM>
    __device__ __forceinline__ void cluster_exchange(double (&sv)[KS > 0 ? KS : 1], double (&mv)[KM > 0 ? KM : 1]) {
        constexpr int K = KS + KM;
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        cl_epoch += 1;
        gu64* gran = (gu64*)cl_part + (size_t)(cl_epoch & 1u) * (kMaxCluster * 16);
        if (tid < 64) {
            int lane = tid;
            asm volatile("" : "+v"(lane));  // (else the lane masks below -- one set per K -- are computed at the kernel's entry and kept in spilled SGPRs)
            if (lane < 2 * K) {  // granule (value k = lane >> 1, half h = lane & 1) of this workgroup
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double vk = k < KS ? sv[k < KS ? k : 0] : mv[k >= KS ? k - KS : 0];
                    v = (lane >> 1) == k ? vk : v;
                }
                const unsigned long long b = (unsigned long long)__double_as_longlong(v);
                const unsigned half = (lane & 1) ? (unsigned)(b >> 32) : (unsigned)(b & 0xffffffffull);
                __hip_atomic_store(gran + crank * 16 + lane, ((unsigned long long)cl_epoch << 32) | half, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            // A member's granules sit in a 128-byte line of their own (16 slots; shared lines made the write-through stores
            // of different members slower), but the sweep visits the live ones only: lane-linear index ql = rank*2K + j
            // reads slot rank*16 + j, so that the common cases -- one value (two-loop recursion, pass barrier) or three --
            // are one 64-lane sweep for clusters of up to 32 / 10 members.
            // (Stencil model only: with the elementwise models the plain slot-linear sweep measured faster, 1.55 against
            // 1.78 ms on noise_1e6, which meets in three reductions per problem; the stencil model meets twice per pass.)
            constexpr bool kLiveOnly = Model::kStencil;
            const int total = kLiveOnly ? csize * 2 * K : csize * 16;
            unsigned spins = 0;
            unsigned long long t_wait0 = 0;
            for (int q0 = 0; q0 < total; q0 += 64) {
                const int ql = q0 + lane;
                const int q = kLiveOnly ? (ql / (2 * K)) * 16 + ql % (2 * K) : ql;
                const bool live = ql < total && (kLiveOnly || (q & 15) < 2 * K);
                unsigned long long gv = 0;
                for (;;) {
                    bool ok = true;
                    if (live) {
                        gv = __hip_atomic_load(gran + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (unsigned)(gv >> 32) == cl_epoch;
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull || cl_aborted) break;
                    // bounded by TIME (s_memrealtime: the constant 100 MHz counter, looked at every 1024 sweeps): members that
                    // are not all resident must not hang the GPU, and a peer that is merely late (a collective kernel beside
                    // the launch, a pre-empted wave) must not be taken for one -- 4 seconds
                    if ((++spins & 0x3ffu) == 0) {
                        unsigned long long now;
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                        if (t_wait0 == 0) t_wait0 = now;
                        else if (now - t_wait0 > 400000000ull) {
                            __hip_atomic_store((gi32*)a.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            cl_aborted = true;
                        }
                    }
                }
                if (live) reinterpret_cast<unsigned*>(exch)[q] = (unsigned)(gv & 0xffffffffull);
            }
        }
        wg_barrier<!Model::kStencil>();   // (the values travel through LDS)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double acc = exch[k];
            for (int c = 1; c < csize; ++c) {
                const double v = exch[c * 8 + k];
                acc = k < KS ? acc + v : __builtin_fmax(acc, v);
            }
            if (k < KS) sv[k < KS ? k : 0] = uniform(acc);
            else mv[k >= KS ? k - KS : 0] = uniform(acc);
        }
    }
    // Workgroup reduction, then (cluster mode) the cluster reduction.  The stencil model's passes read neighbours
    // that other workgroups wrote: it needs the release/acquire form; the elementwise models exchange scalars only.
    template <int KS, int KM>
    __device__ __forceinline__ void reduce(double (&sv)[KS > 0 ? KS : 1], double (&mv)[KM > 0 ? KM : 1]) {
        // stencil model in a cluster: this pass's (write-through) stores have left every wave before the workgroup's
        // barrier inside block_allreduce, i.e. before wave 0 publishes the epoch the other members wait for
        if constexpr (Place::kCluster && Model::kStencil) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        block_allreduce<T, KS, KM, !Model::kStencil>(sv, mv, red, parity, tid);
        if constexpr (Place::kCluster) cluster_exchange<KS, KM>(sv, mv);
    }
    // Orders this pass's vector stores before the next pass's neighbour reads (stencil model).
    __device__ __forceinline__ void pass_barrier() {
        if constexpr (Place::kCluster) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            double z1[1] = {0.0}, z2[1] = {0.0};
            cluster_exchange<1, 0>(z1, z2);
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
        else if constexpr (kBig) return block_of_big((int)a.N, a.ntheta, big_rcpN, i);
        else if constexpr (Place::kResident) {
            // (laundered: the extraction and everything derived from it -- the lane masks k == b of the score sums, 20 slots x
            // MAXB of them, the LDS addresses of the per-block coefficients -- is otherwise hoisted to the kernel's entry, held
            // across the persistent loop and spilled: 172 SGPRs and 21 VGPRs for FunnelModel<4>, 353 / 33 for <8>)
            unsigned w = pk[jj / 10];
            asm volatile("" : "+v"(w));
            return (int)((w >> (3 * (jj % 10))) & 7u);
        }
        else return block_of<MAXB>(a, i);
    }
    __device__ __forceinline__ double ivk(int jj, int i) const {
        if constexpr (MAXB == 1) return iv0;
        else if constexpr (kBig) return big_iv[blk(jj, i)];
        else return a.cur.t.iv[blk(jj, i)];
    }
    __device__ __forceinline__ double sdk(int jj, int i) const {
        if constexpr (MAXB == 1) return sd0;
        else if constexpr (kBig) return big_sd[blk(jj, i)];
        else return sh_sd[blk(jj, i)];
    }
    __device__ __forceinline__ double sdk_at(int i) const {  // element index only (rolled loops)
        if constexpr (MAXB == 1) return sd0;
        else if constexpr (kBig) return big_sd[blk(0, i)];
        else return sh_sd[block_of<MAXB>(a, i)];
    }
    // The coefficients of the element in slot jj / at index i AS THE MODEL'S FUNCTIONS TAKE THEM: the block's exp(-theta) / exp(theta/2)
    // for the one-parameter models (ivk / sdk above), and for the two-parameter family (models.hpp, kPair; include/muse_model.h,
    // MUSE_MODEL_PAIR: block k's parameters are theta[k] and theta[K + k], K = ntheta / 2) the block's four coefficients -- the tables'
    // slots {sd[k], sd[K + k], iv[k], iv[K + k]} of the MAP's theta for the objective, the first two of the theta the problem is
    // DRAWN at (sh_sd) for the draw.  Layout: the tables hold a block's coefficients side by side (models.hpp, pair_table: [block][4];
    // a sampling entry and sh_sd: [block][2]).
    __device__ __forceinline__ auto gcoef(int jj, int i) const {
        if constexpr (Model::kPair && MAXB == 2) {
            // (the zero pad element of an odd-length vector and the phantom slots behind it get ZERO coefficients: a model with a
            //  location parameter has a gradient at x = z = 0, and the header's contract -- no contribution from c = 0, x = z = 0 --
            //  keeps those slots out of every sum)
            const bool valid = i < (int)a.N;
            PairGv c;
#pragma unroll
            for (int q = 0; q < 4; ++q) c.c[q] = valid ? this->pc[q] : 0.0;
            return c;
        } else if constexpr (Model::kPair) {
            PairGp c;   // (models.hpp: the block's record in the LDS copy of the arguments, read by the model's functions themselves)
            c.p = pair_table(a.cur.t) + 4 * blk(jj, i);
            c.valid = i < (int)a.N;
            return c;
        } else {
            return ivk(jj, i);
        }
    }
    __device__ __forceinline__ auto scoef_of_block(int k) const {   // (pair models only)
        PairS c;
        if constexpr (MAXB == 2) {
            c.c[0] = this->psd[0];
            c.c[1] = this->psd[1];
        } else {
            c.c[0] = sh_sd[2 * k];
            c.c[1] = sh_sd[2 * k + 1];
        }
        return c;
    }
    __device__ __forceinline__ auto scoef(int jj, int i) const {
        if constexpr (Model::kPair) return scoef_of_block(blk(jj, i));
        else return sdk(jj, i);
    }
    // The element's share of the block sums the score is assembled from: acc[b] += B(x, z) for the element's block b, or -- two
    // sums per block -- acc[k] += t0, acc[K + k] += t1 (the score's components in the order of theta).
    template <int NB>
    __device__ __forceinline__ void score_add(double* acc, double xi, double zi, int jj, int i, int first = 0) const {
        if constexpr (Model::kPair) {
            double t0, t1;
            Model::score_terms(gcoef(jj, i), xi, zi, t0, t1, i);
            const int k = blk(jj, i) - first, K = a.ntheta >> 1;
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] += (k == b) ? t0 : ((k + K == b) ? t1 : 0.0);
        } else {
            const double t = Model::score_term(xi, zi, i);
            if constexpr (NB == 1) {
                acc[0] += t;
            } else {
                const int k = blk(jj, i) - first;
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] += (k == b) ? t : 0.0;
            }
        }
    }
    __device__ __forceinline__ void pack_blocks() {
        pk[0] = pk[1] = 0u;
        if constexpr (MAXB > 1 && Place::kResident) {
            static_assert(2 * EPT <= 20, "two words of ten 3-bit slots");
#pragma unroll
            for (int jj = 0; jj < 2 * EPT; ++jj) {
                const int i = 2 * (tfirst + (jj >> 1) * ps()) + (jj & 1);
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
    __device__ __forceinline__ static Pair load_pair(const rsrc_t& rs, int i0) {  // the thread's own pair
        Pair p;
        load_f64x2(rs, i0, p.a, p.b);
        return p;
    }
    __device__ __forceinline__ static Pair load_edge(const rsrc_t& rs, int i0) {  // a pair another wave / workgroup owns
        Pair p;
        load_f64x2<Place::kCoherent ? kCoherent : kPlain>(rs, i0, p.a, p.b);
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
                Pair ze = load_edge(z.rsrc, ie);
                const double xe = x.get1(ix);
                Pair sp{0.0, 0.0}, ztp = zp;
                if constexpr (USE_S) {
                    s.own_pair(i0, sp.a, sp.b);
                    const Pair se = load_edge(s.rsrc, ie);
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

    // ---- background generator (streaming clusters of the elementwise models) --------------------------------------
    // A streaming problem is a generator pass (VALU-bound: Philox + Box-Muller + the initial evaluation and first trial,
    // writes x and s = -g) followed by passes that only stream (HBM-bound).  A cluster knows its next problem (problems
    // are dealt round-robin), so the streaming passes of problem p carry the generator pass of problem p + nclusters: every
    // trip first draws kBgPairs pairs of the next problem -- arithmetic only, placed before the trip's loads are used --
    // and stores them with the trip's own stores, into the other (x, s) buffer pair of the cluster's scratch.  The sums
    // of the next problem's initial evaluation accumulate in registers; whatever is left when problem p ends is
    // drawn in the foreground.  A thread draws its pairs in the same order either way: bit-identical results.
    static constexpr bool kBg = Place::kCluster && !Place::kResident && !Model::kStencil;
    static constexpr int kBgPairs = MUSE_BG_PAIRS, kBgU = MUSE_BG_U;
    struct GenState {
        int p;         // the problem being generated (-1: none)
        int trip;      // pairs tfirst + k*pstride, k < trip, are done
        bool active;   // the streaming passes of the current problem still have trips of it to draw
        uint64_t sim;
        double sum[4], mx[2];
    };
    GenState gen;
    int bufsel;        // (x, s) buffer pair of the current problem
    int next_p;        // the cluster's next problem, -1: none
    rsrc_t bg_xr, bg_sr;
    double bgx[2 * kBgPairs], bgs[2 * kBgPairs];
    __device__ __forceinline__ int gen_trips() const { return ((int)a.ld / 2 + pstride - 1) / pstride; }
    // one pair of problem gen.p: elements i0, i0 + 1 (the arithmetic of the fused sampler pass in begin())
    __device__ __forceinline__ void gen_pair(int i0, double* xo, double* so) {
        const int N = (int)a.N;
        const uint64_t idx[2] = {(uint64_t)i0, (uint64_t)(i0 + 1)};
        NormalPair np[2];
        normal_pairs<2>(a.seed, gen.sim, idx, np);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int i = i0 + v;
            const bool valid = i < N;
            double zt, xt;
            Model::sample(scoef(v, i), np[v].n1, np[v].n2, zt, xt, i);
            xt = valid ? xt : 0.0;
            const auto ivi = gcoef(v, i);
            const double gi = Model::grad(ivi, xt, 0.0, gen.sum[0], i);
            const double sd = -gi;
            gen.sum[1] = fma(gi, sd, gen.sum[1]);
            gen.mx[0] = absmax(gen.mx[0], gi);
            const double zt1 = fma(1.0, sd, 0.0);
            const double gt = Model::grad(ivi, xt, zt1, gen.sum[2], i);
            gen.sum[3] = fma(gt, sd, gen.sum[3]);
            gen.mx[1] = absmax(gen.mx[1], gt);
            xo[v] = xt;
            so[v] = sd;
        }
    }
    __device__ __forceinline__ void bg_draw() {  // between a trip's loads and their use: arithmetic only
        if constexpr (kBg) {
            if (__builtin_amdgcn_readfirstlane((int)gen.active)) {
#pragma unroll
                for (int q = 0; q < kBgPairs; ++q) gen_pair(2 * (tfirst + (gen.trip + q) * pstride), bgx + 2 * q, bgs + 2 * q);
            }
        }
    }
    __device__ __forceinline__ void bg_store() {  // with the trip's stores
        if constexpr (kBg) {
            if (__builtin_amdgcn_readfirstlane((int)gen.active)) {
#pragma unroll
                for (int q = 0; q < kBgPairs; ++q) {
                    const int i0 = 2 * (tfirst + (gen.trip + q) * pstride);  // past the end: dropped by the range check
                    store_f64x2(bg_xr, i0, bgx[2 * q], bgx[2 * q + 1]);
                    store_f64x2(bg_sr, i0, bgs[2 * q], bgs[2 * q + 1]);
                }
                gen.trip += kBgPairs;
                gen.active = gen.trip < gen_trips();
            }
        }
    }
    // The element loop of a streaming pass that carries the background generator: a trip issues the loads of the vectors
    // the pass reads (r0, r1, r2: ReadIf entries), draws the next problem's pairs while they are in flight, runs the
    // element bodies, and stores the trip's results and the drawn pairs together.
    template <class R0, class R1, class R2, class F, class... W>
    __device__ __forceinline__ void pass_elems_r(R0 r0, R1 r1, R2 r2, F&& f, W&&... written) {
        if constexpr (kBg) {
            int t = tfirst;
            asm volatile("" : "+v"(t));
            const int n = (int)a.ld, pstr = pstride;
            constexpr int UB = kBgU < U ? kBgU : U;  // shorter trips: fewer loaded registers live across the drawing
#pragma unroll 1
            for (int i0 = 2 * t; i0 < n; i0 += 2 * UB * pstr) {
                r0.template preload<UB>(i0, pstr);
                r1.template preload<UB>(i0, pstr);
                r2.template preload<UB>(i0, pstr);
                bg_draw();
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = i0 + 2 * u * pstr;
                    f(2 * u, i);
                    f(2 * u + 1, i + 1);
                }
                (written.template flush<UB>(i0, pstr), ...);
                bg_store();
                r0.unload();
                r1.unload();
                r2.unload();
            }
        } else {
            for_elems<T, EPT, U>(a.ld, tfirst, ps(), f, written...);
        }
    }

    // -- objective/gradient at z + c s (or at z when !USE_S).  Returns f = -logLike,
    //    dphi = grad . s and gmax = ||grad||_inf; the gradient itself is stored (into g) only when
    //    STORE_G -- a line-search trial needs just the three scalars.  One pass, one barrier.
    // An element pass whose body needs the current z: f(zz, jj, i) with zz a std::bool_constant -- true where z is
    // still the unwritten zero start of the streaming policy (the body then uses 0 and issues no load).
    template <class F, class... W>
    __device__ __forceinline__ void for_elems_zz(F&& f, W&&... written) {
        if constexpr (!Place::kResident && !Model::kStencil) {
            if (z_zero) {
                for_elems<T, EPT, U>(a.ld, tfirst, ps(), [&](int jj, int i) { f(std::true_type{}, jj, i); }, written...);
                return;
            }
        }
        for_elems<T, EPT, U>(a.ld, tfirst, ps(), [&](int jj, int i) { f(std::false_type{}, jj, i); }, written...);
    }
    // ... and the same for a pass that reads z (unless it is the unwritten zero), r1 and r2, with the background generator
    template <class R1, class R2, class F, class... W>
    __device__ __forceinline__ void for_elems_zz_r(R1 r1, R2 r2, F&& f, W&&... written) {
        if constexpr (!Place::kResident && !Model::kStencil) {
            if (z_zero) {
                pass_elems_r(reads(z, false), r1, r2, [&](int jj, int i) { f(std::true_type{}, jj, i); }, written...);
                return;
            }
        }
        pass_elems_r(reads(z), r1, r2, [&](int jj, int i) { f(std::false_type{}, jj, i); }, written...);
    }

    //    INIT_S (resident policy, where s lives in registers): the pass also sets the steepest-descent
    //    direction s = -g and returns dphi = g . s, exactly as the separate pass would.
    //    SPEC (a line-search trial of the elementwise models, kSpec): the pass ALSO forms the sums that the solve's last pass
    //    would form if this trial turns out to be the accepted step and the solve ends with it -- the step's norm and the score
    //    terms at z + c s -- in the same per-thread order and through the same reduction trees (each value's tree is its own),
    //    so that the last pass shrinks to the step and the store of the MAP, without score terms and without a reduction
    //    (solve(), `spec_hit`): the same bits.  On a quadratic objective -- every built-in model -- HagerZhang's secant step is
    //    accepted and ends the solve: of configs[1]'s three evaluation passes with a reduction each, two are left.  In the
    //    streaming placements a trial from the virtual zero start also writes z + c s into the problem's MAP slot (spec_on: the
    //    first L-BFGS iteration's trials), which then IS the last pass's output: a whole pass less over HBM.  A rejected trial's
    //    values are dropped (a slot it wrote is overwritten by the pass that does end the solve; nothing reads it in between).
    // STREAMING placements only (MUSE_SPEC_RESIDENT=1 builds it for the resident ones too).  Measured on one box, stamps and A/B
    // (round 5): in the LDS-resident kernel the last pass is bound by the 80 KB store of the MAP, its arithmetic hides under the
    // stores, and the trial -- whose arithmetic does not hide -- grows by 1.4 k cycles to save a 1.9 k reduction: 35.1 -> 34.5 k
    // cycles per problem in the map kernel, and a LOSS in the loop kernel (24.5 -> 25.6 k: 57 more spilled registers).  In the
    // streaming placements a whole pass over HBM disappears: noise_1e6 1.303 -> 1.048 ms per step.
    // (one component: the per-block select chain of several components' score sums inside the trial spilled; MUSE_SPEC_MAXB)
#ifndef MUSE_SPEC_MAXB
#define MUSE_SPEC_MAXB 1
#endif
#ifndef MUSE_SPEC_RESIDENT
#define MUSE_SPEC_RESIDENT 0
#endif
    static constexpr bool kSpec = !Model::kStencil && !kBig && KB <= MUSE_SPEC_MAXB && (!Place::kResident || MUSE_SPEC_RESIDENT);
    double spec_c;          // the trial the speculative values belong to (NaN: none)
    double spec_acc[KB], spec_mx;
    bool spec_on;           // this line search's trials speculate (the first L-BFGS iteration: where one-iteration solves end)
    template <bool USE_S, bool STORE_G, bool INIT_S = false, bool ZZ = false, bool SPEC = false>
    __device__ __forceinline__ void eval(double c, double& f, double& dphi, double& gmax) {
        constexpr int KX = SPEC ? KB : 0;     // extra sums / maxima of a speculating trial
        double sum[2 + KX] = {0.0, 0.0}, mx[1 + (SPEC ? 1 : 0)] = {0.0};
#pragma unroll
        for (int b = 0; b < KX; ++b) sum[2 + b] = 0.0;
        if constexpr (SPEC) mx[1] = 0.0;
        if constexpr (!Model::kStencil) {
            pass_elems_r(reads(z, !ZZ), reads(s, USE_S), reads(x), [&](int jj, int i) {
                const double zo = ZZ ? 0.0 : z.get(jj, i);  // ZZ: z is still the (unwritten) zero start
                double zi = zo;
                double si = 0.0;
                if constexpr (USE_S) {
                    si = s.get(jj, i);
                    zi = fma(c, si, zi);
                }
                const double xi = x.get(jj, i);
                const double gi = Model::grad(gcoef(jj, i), xi, zi, sum[0], i);
                if constexpr (STORE_G) g.set(jj, i, gi);
                if constexpr (USE_S) sum[1] = fma(gi, si, sum[1]);
                if constexpr (INIT_S) {
                    const double sd = -gi;
                    s.set(jj, i, sd);
                    sum[1] = fma(gi, sd, sum[1]);
                }
                mx[0] = absmax(mx[0], gi);
                if constexpr (SPEC) {   // the sums of solve()'s last pass, should this be the accepted step (same statements, same order)
                    mx[1] = absmax(mx[1], zi - zo);
                    if constexpr (!Place::kResident) z.set(jj, i, zi);   // (streaming, from the virtual zero: the MAP slot itself)
                    score_add<KB>(sum + 2, xi, zi, jj, i);
                }
            }, when(STORE_G, g), when(INIT_S, s), when(SPEC && !Place::kResident && spec_on, z));
        } else {
            stencil_pairs<USE_S>(c, sum[0], [](int, int) {}, [&](int u, int i0, double g0, double g1, double s0, double s1) {
                if constexpr (STORE_G) {
                    g.set(2 * u, i0, g0);
                    g.set(2 * u + 1, i0 + 1, g1);
                }
                if constexpr (USE_S) sum[1] = fma(g1, s1, fma(g0, s0, sum[1]));
                if constexpr (INIT_S) {  // the steepest-descent direction and g . s with the initial evaluation
                    s.set(2 * u, i0, -g0);
                    s.set(2 * u + 1, i0 + 1, -g1);
                    sum[1] = fma(g1, -g1, fma(g0, -g0, sum[1]));
                }
                mx[0] = absmax(absmax(mx[0], g0), g1);
            }, when(STORE_G, g), when(INIT_S, s));
        }
        reduce<2 + KX, 1 + (SPEC ? 1 : 0)>(sum, mx);
        f = 0.5 * (sum[0] + a.cur.f_const);
        dphi = sum[1];
        gmax = nan_if(sum[0] != sum[0] || sum[1] != sum[1], mx[0]);
        f_calls += 1;
        if constexpr (SPEC) {
#pragma unroll
            for (int b = 0; b < KB; ++b) spec_acc[b] = sum[2 + b];
            spec_mx = mx[1];
            // (streaming without the stores: the MAP slot was not written; MUSE_DEBUG bit 5: never a hit -- A/B runs and tests)
            spec_c = ((Place::kResident && !(a.debug & 32)) || spec_on) ? c : __builtin_nan("");
        }
    }

    // The initial evaluation of the resident elementwise layout, fused with the FIRST line-search trial: the
    // steepest-descent direction s = -g is known element by element, so the same pass also evaluates the objective
    // at z + c0 s (InitialStatic: c0 = 1) -- what the line search would ask for first -- and one 6-value reduction
    // replaces two passes with a 3-value reduction each.  The trial's scalars wait in trial_*; the line search
    // consumes them (and counts the evaluation) only if it gets that far.
    double trial_c, trial_phi, trial_dphi, trial_gmax;
    bool have_trial;
    // Streaming policy, elementwise models, start from zero(z) or the true z: begin() has already made the initial
    // evaluation and the first trial inside the sampler pass (the values were in registers there), and with a
    // zero start z is not written at all until the first update ("z_zero": passes take z = 0 instead of reading it).
    bool init_done, z_zero;
    double init_f, init_dphi, init_gmax;
    __device__ __forceinline__ void eval_init_with_trial(double c0, double& f, double& dphi, double& gmax) {
        double sum[4] = {0.0, 0.0, 0.0, 0.0}, mx[2] = {0.0, 0.0};
        for_elems<T, EPT, U>(a.ld, tfirst, ps(), [&](int jj, int i) {
            const double zi = z.get(jj, i), xi = x.get(jj, i);
            const auto ivi = gcoef(jj, i);
            const double gi = Model::grad(ivi, xi, zi, sum[0], i);
            const double sd = -gi;
            s.set(jj, i, sd);
            sum[1] = fma(gi, sd, sum[1]);
            mx[0] = absmax(mx[0], gi);
            const double zt = fma(c0, sd, zi);  // exactly the trial point eval<true, false>(c0) would form
            const double gt = Model::grad(ivi, xi, zt, sum[2], i);
            sum[3] = fma(gt, sd, sum[3]);
            mx[1] = absmax(mx[1], gt);
        }, s);
        reduce<4, 2>(sum, mx);
        f = 0.5 * (sum[0] + a.cur.f_const);
        dphi = sum[1];
        gmax = nan_if(sum[0] != sum[0] || sum[1] != sum[1], mx[0]);
        f_calls += 1;
        trial_c = c0;
        trial_phi = 0.5 * (sum[2] + a.cur.f_const);
        trial_dphi = sum[3];
        trial_gmax = nan_if(sum[2] != sum[2] || sum[3] != sum[3], mx[1]);
        have_trial = true;
    }

    // phi(c), dphi(c) of the line search (the NLSolversBase objective cache is last_c/last_phi).
    __device__ __forceinline__ void phidphi(double c, double& phi, double& dphi) {
        if (have_trial) {
            have_trial = false;
            if (c == trial_c) {  // the first trial of the first line search: evaluated with the initial point
                phi = trial_phi;
                dphi = trial_dphi;
                last_c = c;
                last_gmax = trial_gmax;
                last_phi = phi;
                f_calls += 1;
                return;
            }
        }
        double gm;
        if constexpr (!Place::kResident && !Model::kStencil) {
            // (ONE instantiation of the trial per case: a speculating and a plain one side by side at the line search's evaluation
            //  site cost the LDS-resident kernel 226 spilled registers; spec_on only gates the speculative STORES)
            if (z_zero) eval<true, false, false, true, kSpec>(c, phi, dphi, gm);
            else eval<true, false>(c, phi, dphi, gm);
        } else if constexpr (kSpec) {
            eval<true, false, false, false, true>(c, phi, dphi, gm);
        } else {
            eval<true, false>(c, phi, dphi, gm);
        }
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
        // the constants are materialised here (two moves each): hoisted to the kernel's entry they stay live across
        // the persistent loop and, in the register-bound placements, come back from scratch in every line search
        double delta = kHzDelta, sigma = kHzSigma, delta2 = 2.0 * kHzDelta - 1.0;
        if constexpr (Place::kResident) asm volatile("" : "+v"(delta), "+v"(sigma), "+v"(delta2));
        const bool w1 = (delta * dphi_0 >= (phi_c - phi_0) / c) && (dphi_c >= sigma * dphi_0);
        const bool w2 = (delta2 * dphi_0 >= dphi_c) && (dphi_c >= sigma * dphi_0) && (phi_c <= phi_lim);
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
    __device__ __forceinline__ bool linesearch(double c0, double phi_0, double dphi_0, double& alpha) {
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

    // (run / begin / solve / finish / linesearch are ALWAYS inlined: left to the inliner's budget, the library build -- a hundred
    // instantiations in one translation unit -- kept Solver::run as a function of its own for some of them (FunnelModel<8> in every
    // placement, the non-cluster streaming placements of every model: two call sites, the map kernel's and the loop kernel's), and
    // the solver's state, its register-resident vectors included, then lived in memory behind `this`: FunnelModel<8> at
    // N = 10^4 ran 189 us per 512-sim step against 67 us for FunnelModel<4>.  A single-instantiation build (tools/regs.py
    // --check) does not show it: tools/regs.py --library reads the product's own code object.)
    __device__ __forceinline__ void run(int p, double* wg_scratch, double* lds_x, double* lds_g) {
        Prefetch<EPT> none;
        none.p = -1;
        none.g_pending = false;
        run(p, wg_scratch, lds_x, lds_g, none);
    }
    // (next_p: the workgroup's next problem of the same iteration, -1: none -- its n1 travels into the g area during this solve)
    // (FD: the launch may be a finite-difference map that carries its fiducial -- never the loop kernel's, which says so: its code
    //  then holds neither the wait nor the publication)
    template <bool FD = true>
    __device__ __forceinline__ void run(int p, double* wg_scratch, double* lds_x, double* lds_g, Prefetch<EPT>& pf, int next_p = -1) {
        begin<false, FD>(p, wg_scratch, lds_x, lds_g, pf);
        if (d.normals_only) return;  // the element only filled its slot of the normals cache
        pfp = &pf;
        if constexpr (Place::kResident && Place::kXgLds) {
            if (next_p >= 0 && a.ncache_mode == 2 && !(a.debug & 4)) prefetch_issue<T>(a, tid, next_p, lds_x, lds_g, pf, false);
        }
        solve(p);
        finish(p);
        if constexpr (FD) {
            if (is_fid(p)) publish_fiducial();
        }
    }
    // ---- a finite-difference launch that carries its own fiducial MAP (BatchArgs::fd_fold; src/muse.jl:417-432 in ONE launch) ----
    // The fiducial's workgroup: every wave's stores of the MAP have been acknowledged by the L2 (vmcnt(0)), the workgroup meets, and
    // one thread's agent-scope RELEASE store of the launch's tag writes the L2's dirty lines back before the flag can be seen -- the
    // compute units of the other XCDs read the MAP through L2s of their own.  ONE such event per launch: a fence is the right tool
    // here (the per-pass exchanges of the cluster placements use tagged granules instead, which cost no write-back).
    // (derived from the launch's arguments where they are needed, not kept in the problem's descriptor: that lives in registers
    //  through the solve)
    __device__ __forceinline__ bool is_fid(int p) const { return a.kind == BATCH_FD && a.fd_fold && p == 0; }
    __device__ __forceinline__ bool wait_fid(int p) const { return a.kind == BATCH_FD && a.fd_fold && p != 0; }
    __device__ __forceinline__ int info_row(int p) const {   // a problem's solver info: its index, but for a launch that carries its fiducial
        return (a.kind == BATCH_FD && a.fd_fold) ? (p == 0 ? a.nproblems - 1 : p - 1) : p;
    }
    __device__ __forceinline__ void publish_fiducial() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.fid_flag, a.fid_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    // A perturbed problem: its x is staged; before the warm start is loaded the flag must carry this launch's tag.  Once per
    // WORKGROUP (the outcome is remembered in LDS beside the ticket): the lines of the fiducial's slot cannot be in this compute
    // unit's L1 or this XCD's L2 from earlier in the launch -- nobody reads the slot before the flag is seen, and the launch began with
    // the caches invalidated -- but the acquire fence behind the poll is what the memory model asks for, and once per workgroup it is
    // free.  Bounded by time (2 s) like every other wait of the engine: an expired wait raises the launch's error word.
    __device__ __forceinline__ void wait_fiducial() {
        int* seen = reinterpret_cast<int*>(sh_rho + 40) + 2;   // ticket[2] of the kernel's LDS carve (kernels.hpp)
        if ((unsigned)__builtin_amdgcn_readfirstlane(seen[0]) == a.fid_tag) return;
        // ONE wavefront polls (every workgroup of the launch but one waits here for ~20 us: eight polling wavefronts each were a stream
        // of requests for one address that the fiducial's own traffic had to queue behind); the others wait at the barrier
        if ((__builtin_amdgcn_readfirstlane(tid) >> 6) == 0) {
            unsigned long long t0 = 0;
            unsigned spins = 0;
            while (__hip_atomic_load(a.fid_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.fid_tag) {
                __builtin_amdgcn_s_sleep(32);
                if ((++spins & 0x3fu) == 0) {
                    unsigned long long now;
                    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > 200000000ull) {
                        typedef __attribute__((address_space(1))) int gi32;
                        __hip_atomic_store((gi32*)a.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid == 0) seen[0] = (int)a.fid_tag;
    }
    // The loop kernel's first problem of an outer iteration > 1 (LDS-resident layout): the problem this workgroup solved LAST in
    // the previous iteration (it visits its elements in alternating order), so its warm start -- the MAP that solve ended with --
    // is still in the z registers, which the caller carried over: nothing is cleared or loaded for z.  The registers hold exactly
    // what that solve's last pass stored into the problem's slot (phantom slots: zeros, as an out-of-range load returns), so the
    // results are the same bits.  A path of its own, beside begin() instead of inside it: with z live through ALL of begin() --
    // the generator loop, the staging of a true-z start -- the register allocator spilled it (40 -> 111 spilled registers).
#ifndef MUSE_KEEPZ_MAXB
#define MUSE_KEEPZ_MAXB 8   // (every tier: with ONE copy of the solve in the loop kernel the kept MAP costs no spilled register; 1 = one component only)
#endif
    static constexpr bool kKeepZ = Place::kResident && Place::kXgLds && !Model::kStencil && MAXB <= MUSE_KEEPZ_MAXB;
    // (workgroup-uniform) can problem p start from kept registers: a warm start whose x comes from cached normals or from the data
    __device__ __forceinline__ bool can_keep(int p) const {
        if constexpr (!kKeepZ) return false;
        const ProblemDesc dd = describe(a, p);
        return dd.z0_mode == Z0_WARM && !dd.normals_only && dd.tsample < 0 &&
               ((dd.x_mode == X_SAMPLE && dd.nslot >= 0 && a.ncache_mode == 2) || dd.x_mode == X_DATA);
    }
    // (workgroup-uniform) will problem p, solved LAST in this iteration of the device-resident loop, start the next iteration from the
    // registers it ends in (can_keep as the next iteration will see it: a warm start, the normals from the cache)?  Then its MAP need not
    // go to memory now: nothing reads the slot before the loop ends, and the worker stores it then (store_kept).
    __device__ __forceinline__ bool keeps_next(int p) const {
        if constexpr (!kKeepZ) return false;
        const ProblemDesc dd = describe(a, p);
        return !dd.normals_only && dd.tsample < 0 && dd.zslot >= 0 &&
               ((dd.x_mode == X_SAMPLE && dd.nslot >= 0 && a.ncache != nullptr && a.ncache_mode != 0) || dd.x_mode == X_DATA);
    }
    __device__ __forceinline__ void store_kept(int p) {   // z (registers) -> problem p's slot
        if constexpr (kKeepZ) {
            const ProblemDesc dd = describe(a, p);
            if (dd.zslot < 0) return;
            const int64_t ld = a.ld;
            VH zout;
            zout.bind(a.zhat + dd.zslot * ld, ld);
            for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) { zout.set(jj, i, z.get(jj, i)); });
        }
    }
    // what run() does behind begin() (the loop kernel picks the begin itself: kept MAP or not)
    __device__ __forceinline__ void after_begin(int p, double* lds_x, double* lds_g, Prefetch<EPT>& pf, int next_p) {
        if (d.normals_only) return;
        pfp = &pf;
        if constexpr (Place::kResident && Place::kXgLds) {
            if (next_p >= 0 && a.ncache_mode == 2 && !(a.debug & 4)) prefetch_issue<T>(a, tid, next_p, lds_x, lds_g, pf, false);
        }
        solve(p);
        finish(p);
    }
    __device__ __forceinline__ void begin_kept(int p, double* wg_scratch, double* lds_x, double* lds_g, Prefetch<EPT>& pf) {
        if constexpr (kKeepZ) {
            d = describe(a, p);
            const bool pf_hit = pf.p == p;
            const int64_t N = a.N, ld = a.ld;
            init_done = false;
            z_zero = false;
            have_trial = false;
            stamp(p, 0);
            iv0 = a.cur.t.iv[0];
            sd0 = a.cur.t.sd[0];
            if constexpr (Model::kPair && MAXB == 2) {
                this->pc[0] = this->psd[0] = a.cur.t.sd[0];   // (one block: pair_table's record 0 = sd[0..3])
                this->pc[1] = this->psd[1] = a.cur.t.sd[1];
                this->pc[2] = a.cur.t.sd[2];
                this->pc[3] = a.cur.t.sd[3];
            }
            if constexpr (MAXB > 1) {
                int tl = tid;
                asm volatile("" : "+v"(tl));
                if constexpr (Model::kPair) {   // the draw's two coefficients per block, [block][2], out of the [block][4] table
                    if (tl < MAXB) sh_sd[tl] = pair_table(a.cur.t)[4 * (tl >> 1) + (tl & 1)];
                } else {
                if (tl < MAXB) sh_sd[tl] = a.cur.t.sd[tl];
                }
                wg_barrier<true>();
            }
            hist = wg_scratch;
            x.bind(lds_x, ld);
            g.bind(lds_g, ld);
            if (d.x_mode == X_SAMPLE) {
                // (begin()'s cached-normals path without its warm-start loads)
                const rsrc_t n1r = make_rsrc(a.ncache + (int64_t)(2 * d.nslot) * ld, ld * 8);
                const rsrc_t n2r = make_rsrc(a.ncache + (int64_t)(2 * d.nslot + 1) * ld, ld * 8);
                int tl = tid;
                asm volatile("" : "+v"(tl));
                const bool n1_here = pf_hit && pf.have_n1, n2_here = pf_hit && pf.have_n2;
                if (n1_here && n2_here) {
                    // both vectors are in LDS (the usual case: fetched while the workgroup waited for theta): pair by pair, in
                    // place -- nothing but the pair in hand is live beside the kept z
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA loads have landed
#pragma unroll
                    for (int j = 0; j < EPT; ++j) {
                        const int i0 = 2 * (tl + j * T);
                        const double a0 = g.get(2 * j, i0), a1 = g.get(2 * j + 1, i0 + 1);
                        const double b0 = x.get(2 * j, i0), b1 = x.get(2 * j + 1, i0 + 1);
                        double zt0, xt0, zt1, xt1;
                        Model::sample(scoef(2 * j, i0), a0, b0, zt0, xt0, i0);
                        Model::sample(scoef(2 * j + 1, i0 + 1), a1, b1, zt1, xt1, i0 + 1);
                        const bool valid1 = i0 + 1 < (int)N;
                        if constexpr (Model::kPair) xt0 = i0 < (int)N ? xt0 : 0.0;   // (a location parameter in a slot beyond the vector: begin())
                        x.set(2 * j, i0, xt0);
                        x.set(2 * j + 1, i0 + 1, valid1 ? xt1 : 0.0);
                    }
                } else {
                    double c1[EPT][2], c2[EPT][2];
                    if (n1_here) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (!n2_here) {
#pragma unroll
                        for (int j = 0; j < EPT; ++j) load_f64x2<kNormalsAux>(n2r, 2 * (tl + j * T), c2[j][0], c2[j][1]);
                    }
                    if (!n1_here) {
#pragma unroll
                        for (int j = 0; j < EPT; ++j) load_f64x2<kNormalsAux>(n1r, 2 * (tl + j * T), c1[j][0], c1[j][1]);
                    }
                    if (n1_here) {
#pragma unroll
                        for (int j = 0; j < EPT; ++j) {
                            const int i0 = 2 * (tl + j * T);
                            c1[j][0] = g.get(2 * j, i0); c1[j][1] = g.get(2 * j + 1, i0 + 1);
                        }
                    }
                    if (n2_here) {
#pragma unroll
                        for (int j = 0; j < EPT; ++j) {
                            const int i0 = 2 * (tl + j * T);
                            c2[j][0] = x.get(2 * j, i0); c2[j][1] = x.get(2 * j + 1, i0 + 1);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < EPT; ++j) {
                        const int i0 = 2 * (tl + j * T);
                        double zt0, xt0, zt1, xt1;
                        Model::sample(scoef(2 * j, i0), c1[j][0], c2[j][0], zt0, xt0, i0);
                        Model::sample(scoef(2 * j + 1, i0 + 1), c1[j][1], c2[j][1], zt1, xt1, i0 + 1);
                        const bool valid1 = i0 + 1 < (int)N;
                        if constexpr (Model::kPair) xt0 = i0 < (int)N ? xt0 : 0.0;
                        x.set(2 * j, i0, xt0);
                        x.set(2 * j + 1, i0 + 1, valid1 ? xt1 : 0.0);
                    }
                }
            } else {   // the data element
                if (pf_hit && pf.have_x) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    VH xs;
                    xs.bind(a.x_data, ld);
                    for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) { x.set(jj, i, xs.get(jj, i)); }, x);
                }
            }
            s.clear();
            if (pf_hit) {   // consumed
                pf.p = -1;
                pf.g_pending = false;
            }
            stamp(p, 9);
            wg_barrier<true>();
            stamp(p, 1);
        }
    }
    Prefetch<EPT>* pfp;
    // the solve is about to write g (a kept L-BFGS update): a prefetch that is on its way into the g area must land first --
    // it would overwrite the gradient -- and is given up (the next problem loads its n1 the ordinary way)
    __device__ __forceinline__ void drop_g_prefetch() {
        if constexpr (Place::kResident && Place::kXgLds) {
            if (pfp->g_pending) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pfp->g_pending = false;
                pfp->p = -1;
            }
        }
    }

    // A finite-difference batch's sampling entries (exp(theta/2) per block of the theta a problem is DRAWN at): a device buffer, or
    // -- a few entries shared by the simulations, round 5 -- the kernel-argument segment itself, in the place of maps[] (such a
    // launch carries one map): no upload between the fiducial launch and this one
    __device__ __forceinline__ const SampleSd* tsample_base() const {
        if (a.tsample) return a.tsample;
        return reinterpret_cast<const SampleSd*>(reinterpret_cast<const char*>((const void*)__builtin_amdgcn_kernarg_segment_ptr()) +
                                                 offsetof(BatchArgs, maps));
    }
    // -- phase 1: bind storage, produce x and the starting point
    template <bool KEEP_ZTRUE>
    __device__ __forceinline__ void begin(int p, double* wg_scratch, double* lds_x, double* lds_g) {
        Prefetch<EPT> none;
        none.p = -1;
        none.g_pending = false;
        pfp = nullptr;
        begin<KEEP_ZTRUE>(p, wg_scratch, lds_x, lds_g, none);
    }
    template <bool KEEP_ZTRUE, bool FD = true>
    __device__ __forceinline__ void begin(int p, double* wg_scratch, double* lds_x, double* lds_g, Prefetch<EPT>& pf) {
        d = describe(a, p);
        // the loop kernel fetched this problem's theta-free inputs ahead (LDS-resident layout only; workgroup-uniform)
        const bool pf_hit = Place::kResident && Place::kXgLds && pf.p == p;
        const int64_t N = a.N, ld = a.ld;
        init_done = false;
        z_zero = false;
        have_trial = false;
        stamp(p, 0);
        iv0 = a.cur.t.iv[0];
        sd0 = d.tsample >= 0 ? tsample_base()[d.tsample].sd[0] : a.cur.t.sd[0];
        if constexpr (Model::kPair && MAXB == 2) {   // one block: its coefficients are workgroup-uniform
            this->pc[0] = a.cur.t.sd[0];   // (pair_table's record 0 = sd[0..3])
            this->pc[1] = a.cur.t.sd[1];
            this->pc[2] = a.cur.t.sd[2];
            this->pc[3] = a.cur.t.sd[3];
            this->psd[0] = sd0;
            this->psd[1] = d.tsample >= 0 ? tsample_base()[d.tsample].sd[1] : a.cur.t.sd[1];
        }
        if constexpr (kBig) {
            // (generic pointers into the kernarg segment: BatchArgs is the kernel's only parameter)
            const BigTheta* bt = reinterpret_cast<const BigTheta*>(
                reinterpret_cast<const char*>((const void*)__builtin_amdgcn_kernarg_segment_ptr()) + offsetof(BatchArgs, big));
            big_iv = bt->iv;
            big_sd = d.tsample >= 0 ? reinterpret_cast<const double*>(a.tsample) + (int64_t)d.tsample * kBigTheta : bt->sd;
            big_rcpN = 1.0 / (double)a.N;
        } else if constexpr (MAXB > 1) {
            // FD batches sample at a theta that differs from the MAP theta
            int tl = tid;
            asm volatile("" : "+v"(tl));  // (else tid * 8 is formed at the kernel's entry and held -- spilled -- across it)
            if constexpr (Model::kPair) {   // (a sampling entry is [block][2] already; the MAP's table is [block][4])
                if (tl < MAXB) sh_sd[tl] = d.tsample >= 0 ? tsample_base()[d.tsample].sd[tl] : pair_table(a.cur.t)[4 * (tl >> 1) + (tl & 1)];
            } else {
            if (tl < MAXB) sh_sd[tl] = d.tsample >= 0 ? tsample_base()[d.tsample].sd[tl] : a.cur.t.sd[tl];
            }
            wg_barrier<!Model::kStencil>();
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
            // (x, s) of the cluster's elementwise problems alternate between two buffer pairs: the other one receives
            // the next problem while this one is being solved (background generator)
            double* xbuf = wg_scratch;
            double* sbuf = wg_scratch + 2 * ld;
            if constexpr (kBg) {
                if (bufsel) {
                    xbuf = wg_scratch + (int64_t)(4 + 2 * kM + 1) * ld;
                    sbuf = xbuf + ld;
                }
            }
            x.bind(xbuf, ld);
            g.bind(wg_scratch + ld, ld);
            if constexpr (Place::kLdsS) s.bind(wg_scratch + 2 * ld, ld, lds_x, tfirst, pstride, tid, (int)N);
            else s.bind(sbuf, ld);
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
                const bool z_warm = d.z0_mode != Z0_ZERO && d.z0_mode != Z0_TRUE;
                bool z_loaded = false;   // the start is in zw already
                double zw[EPT][2];
                if (nmode == 2) {
                    // the stream was drawn earlier (in this host call or, for a map the context has seen before, an earlier
                    // one): its normals come from HBM, not from the generator -- all of the thread's loads in flight at
                    // once, the warm start's first (out-of-range pairs read zeros) -- or are here already (Prefetch)
                    double c1[EPT][2], c2[EPT][2];
                    int tl = tid;
                    asm volatile("" : "+v"(tl));  // per-slot offsets recomputed here, not held across the kernel
                    const bool n1_here = pf_hit && pf.have_n1, n2_here = pf_hit && pf.have_n2;   // in the g / x area already
                    if (n1_here) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA loads have landed
                    // (a finite-difference launch that carries its own fiducial MAP: x first -- it does not need the fiducial --, the
                    //  warm start once the fiducial's workgroup has published it: wait_fiducial below)
                    const bool fold_wait = FD && wait_fid(p);   // workgroup-uniform
                    if (z_warm && !fold_wait) {   // the warm start's loads first, then what is not here yet: all in flight at once
#pragma unroll
                        for (int j = 0; j < EPT; ++j) load_f64x2(z0src.rsrc, 2 * (tl + j * T), zw[j][0], zw[j][1]);
                        z_loaded = true;
                    }
                    if (!n2_here) {
#pragma unroll
                        for (int j = 0; j < EPT; ++j) load_f64x2<kNormalsAux>(n2r, 2 * (tl + j * T), c2[j][0], c2[j][1]);
                    }
                    if (!n1_here) {
#pragma unroll
                        for (int j = 0; j < EPT; ++j) load_f64x2<kNormalsAux>(n1r, 2 * (tl + j * T), c1[j][0], c1[j][1]);
                    }
                    if (n1_here) {   // each pair at the slot of the thread that owns it
#pragma unroll
                        for (int j = 0; j < EPT; ++j) {
                            const int i0 = 2 * (tl + j * T);
                            c1[j][0] = g.get(2 * j, i0); c1[j][1] = g.get(2 * j + 1, i0 + 1);
                        }
                    }
                    if (n2_here) {
#pragma unroll
                        for (int j = 0; j < EPT; ++j) {
                            const int i0 = 2 * (tl + j * T);
                            c2[j][0] = x.get(2 * j, i0); c2[j][1] = x.get(2 * j + 1, i0 + 1);
                        }
                    }
                    const bool stage_ztrue = d.z0_mode == Z0_TRUE;   // (only that start reads the true z back from the g area)
#pragma unroll
                    for (int j = 0; j < EPT; ++j) {
                        const int i0 = 2 * (tl + j * T);
                        double zt0, xt0, zt1, xt1;
                        Model::sample(scoef(2 * j, i0), c1[j][0], c2[j][0], zt0, xt0, i0);
                        Model::sample(scoef(2 * j + 1, i0 + 1), c1[j][1], c2[j][1], zt1, xt1, i0 + 1);
                        const bool valid1 = i0 + 1 < (int)N;
                        if constexpr (Model::kPair) {
                            // (a thread's slots beyond the vector -- their normals load as zeros -- draw (0, 0) in the one-parameter family
                            //  by themselves; a location parameter would put its value there)
                            const bool valid0 = i0 < (int)N;
                            xt0 = valid0 ? xt0 : 0.0;
                            zt0 = valid0 ? zt0 : 0.0;
                        }
                        x.set(2 * j, i0, xt0);
                        x.set(2 * j + 1, i0 + 1, valid1 ? xt1 : 0.0);
                        if (stage_ztrue) {
                            g.set(2 * j, i0, zt0);
                            g.set(2 * j + 1, i0 + 1, valid1 ? zt1 : 0.0);
                        }
                    }
                    if (z_warm && fold_wait) {
                        wait_fiducial();
#pragma unroll
                        for (int j = 0; j < EPT; ++j) load_f64x2(z0src.rsrc, 2 * (tl + j * T), zw[j][0], zw[j][1]);
                        z_loaded = true;
                    }
                } else {
                    // sampling sd of the element in (run-time) slot jj: the slot's packed block index selects the value in LDS.
                    // (A select chain over per-block scalars -- boundaries and sd values in SGPRs -- was turned into a
                    // lane-indexed table in SCRATCH by the compiler: a scratch load per element in the generator loop.)
                    auto sd_of = [&](int jj) {
                        if constexpr (MAXB == 1) return sd0;
                        else {
                            const unsigned w = jj >= 10 ? pk[1] : pk[0];
                            const int sh = 3 * (jj >= 10 ? jj - 10 : jj);
                            if constexpr (Model::kPair) return scoef_of_block((int)((w >> sh) & 7u));
                            else return sh_sd[(w >> sh) & 7u];
                        }
                    };
                    // One trip draws kSamplerPairs pairs (2 * kSamplerPairs independent Philox/Box-Muller chains in one basic
                    // block).  At the two waves per SIMD of this placement a wave issues a VALU instruction every ~5.3 cycles
                    // when it has four independent ones to choose from and every ~11 when each depends on the one before
                    // (tools/clockprobe.hip): with one pair per trip the generator ran at ~9.8 cycles per instruction.
                    auto draw = [&](auto npairs, int i0, int j) {
                        // both elements of a pair unconditionally; for odd N the last pair's second element is the pad
                        // slot, kept at 0.  All 2 * P generator chains advance side by side (rng.hpp).
                        constexpr int P = decltype(npairs)::value;
                        uint64_t idx[2 * P];
                        NormalPair np[2 * P];
#pragma unroll
                        for (int q = 0; q < P; ++q) {
                            idx[2 * q] = (uint64_t)(i0 + 2 * T * q);
                            idx[2 * q + 1] = (uint64_t)(i0 + 2 * T * q + 1);
                        }
                        normal_pairs<2 * P>(a.seed, sim, idx, np);
#pragma unroll
                        for (int q = 0; q < P; ++q) {
                            const int j0 = i0 + 2 * T * q;
                            if (nmode == 1) {
                                store_f64x2(n1r, j0, np[2 * q].n1, np[2 * q + 1].n1);
                                store_f64x2(n2r, j0, np[2 * q].n2, np[2 * q + 1].n2);
                            }
                            double zt0, xt0, zt1, xt1;
                            Model::sample(sd_of(2 * (j + q)), np[2 * q].n1, np[2 * q].n2, zt0, xt0, j0);
                            Model::sample(sd_of(2 * (j + q) + 1), np[2 * q + 1].n1, np[2 * q + 1].n2, zt1, xt1, j0 + 1);
                            const bool valid1 = j0 + 1 < (int)N;
                            x.p[j0] = xt0;
                            g.p[j0] = zt0;
                            x.p[j0 + 1] = valid1 ? xt1 : 0.0;
                            g.p[j0 + 1] = valid1 ? zt1 : 0.0;
                        }
                    };
                    int j = 0;  // the pair's slot: pair j of this thread is elements i0, i0 + 1 with i0 = 2 (tid + j T)
                    int tl = tid;
                    asm volatile("" : "+v"(tl));  // (else the first Philox round's product with tid is held -- spilled -- across the kernel)
#pragma unroll 1
                    for (int i0 = 2 * tl; i0 < (int)N; i0 += 2 * T * kSamplerPairs, j += kSamplerPairs) {
                        if constexpr (kSamplerPairs == 2) {
                            if (i0 + 2 * T < (int)N) draw(std::integral_constant<int, 2>{}, i0, j);
                            else draw(std::integral_constant<int, 1>{}, i0, j);
                        } else {
                            draw(std::integral_constant<int, 1>{}, i0, j);
                        }
                    }
                }
                z.clear();
                s.clear();
                if (z_loaded) {
#pragma unroll
                    for (int j = 0; j < EPT; ++j) {
                        z.set(2 * j, 0, zw[j][0]);
                        z.set(2 * j + 1, 0, zw[j][1]);
                    }
                } else {
                    for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                        if (d.z0_mode == Z0_ZERO) z.set(jj, i, 0.0);
                        else if (d.z0_mode == Z0_TRUE) z.set(jj, i, g.get(jj, i));
                        else z.set(jj, i, z0src.get(jj, i));
                    });
                }
            } else {
                x.clear(); g.clear(); z.clear(); s.clear();
                VH ztrue;
                if constexpr (KEEP_ZTRUE) ztrue.bind(extra, ld);
                // streaming: ONE pair per trip (two Philox/Box-Muller chains in flight, as in the rolled
                // resident sampler) -- the pass is generator-bound and loads nothing; a warm start is
                // copied by a pass of its own below
                constexpr int US = Place::kResident ? U : 1;
                const bool z_from_sample = d.z0_mode == Z0_ZERO || d.z0_mode == Z0_TRUE;
                if constexpr (kBg && !KEEP_ZTRUE) {
                    if (d.z0_mode == Z0_ZERO && !d.normals_only && d.tsample < 0) {
                        // the pass described below, through the generator state: the background generator may have drawn
                        // part (or all) of this problem during the previous one's streaming passes
                        if (gen.p != p) {
                            gen.p = p;
                            gen.trip = 0;
                            gen.sim = sim;
#pragma unroll
                            for (int k = 0; k < 4; ++k) gen.sum[k] = 0.0;
                            gen.mx[0] = gen.mx[1] = 0.0;
                        }
                        gen.active = false;
                        const int ntrips = gen_trips();
#pragma unroll 1
                        for (; gen.trip < ntrips; ++gen.trip) {
                            const int i0 = 2 * (tfirst + gen.trip * pstride);
                            double xo[2], so[2];
                            gen_pair(i0, xo, so);
                            store_f64x2(x.rsrc, i0, xo[0], xo[1]);
                            store_f64x2(s.rsrc, i0, so[0], so[1]);
                        }
                        double sum[4] = {gen.sum[0], gen.sum[1], gen.sum[2], gen.sum[3]}, mx[2] = {gen.mx[0], gen.mx[1]};
                        reduce<4, 2>(sum, mx);
                        init_f = 0.5 * (sum[0] + a.cur.f_const);
                        init_dphi = sum[1];
                        init_gmax = nan_if(sum[0] != sum[0] || sum[1] != sum[1], mx[0]);
                        trial_c = 1.0;
                        trial_phi = 0.5 * (sum[2] + a.cur.f_const);
                        trial_dphi = sum[3];
                        trial_gmax = nan_if(sum[2] != sum[2] || sum[3] != sum[3], mx[1]);
                        init_done = true;
                        z_zero = true;
                    }
                }
                if constexpr (!Model::kStencil) {  // (the LDS-resident layout has its own rolled sampler above)
                    if (!init_done && z_from_sample && !d.normals_only) {
                        // sampler + initial evaluation + first line-search trial in ONE pass over values that are in
                        // registers anyway (see eval_init_with_trial for the arithmetic, which is identical): the pass
                        // writes x and s = -g only (and z when the start is the true z) -- against write x, z; read
                        // x, z, write g; read g, write s; read z, s, x of the four passes it replaces
                        const bool ztrue_start = d.z0_mode == Z0_TRUE;
                        double sum[4] = {0.0, 0.0, 0.0, 0.0}, mx[2] = {0.0, 0.0};
                        // the normals of a trip's kGenU pairs are drawn side by side (2 * kGenU generator chains, rng.hpp)
                        constexpr int kGenU = Place::kResident ? 1 : kStreamGenU;
                        NormalPair npv[2 * kGenU];
                        for_elems_pre<T, EPT, kGenU>(ld, tfirst, ps(), [&](auto ucount, int i0, int pstr) {
                            constexpr int UU = decltype(ucount)::value;
                            uint64_t idx[2 * UU];
                            NormalPair got[2 * UU];
#pragma unroll
                            for (int u = 0; u < UU; ++u) {
                                idx[2 * u] = (uint64_t)(i0 + 2 * u * pstr);
                                idx[2 * u + 1] = (uint64_t)(i0 + 2 * u * pstr + 1);
                            }
                            normal_pairs<2 * UU>(a.seed, sim, idx, got);
#pragma unroll
                            for (int q = 0; q < 2 * UU; ++q) npv[q] = got[q];
                        }, [&](int jj, int i) {
                            const bool valid = i < N;
                            const NormalPair np = npv[jj % (2 * kGenU)];
                            double zt, xt;
                            Model::sample(scoef(jj, i), np.n1, np.n2, zt, xt, i);
                            zt = valid ? zt : 0.0;
                            xt = valid ? xt : 0.0;
                            if constexpr (KEEP_ZTRUE) ztrue.set(jj, i, keep_value(valid, zt, sdk(jj, i), np, i));
                            x.set(jj, i, xt);
                            const double z0v = ztrue_start ? zt : 0.0;
                            if (Place::kResident || ztrue_start) z.set(jj, i, z0v);  // registers: always defined
                            const auto ivi = gcoef(jj, i);
                            const double gi = Model::grad(ivi, xt, z0v, sum[0], i);
                            const double sd = -gi;
                            s.set(jj, i, sd);
                            sum[1] = fma(gi, sd, sum[1]);
                            mx[0] = absmax(mx[0], gi);
                            const double zt1 = fma(1.0, sd, z0v);
                            const double gt = Model::grad(ivi, xt, zt1, sum[2], i);
                            sum[3] = fma(gt, sd, sum[3]);
                            mx[1] = absmax(mx[1], gt);
                        }, when(KEEP_ZTRUE, ztrue), x, s, when(Place::kResident || ztrue_start, z));
                        reduce<4, 2>(sum, mx);
                        init_f = 0.5 * (sum[0] + a.cur.f_const);
                        init_dphi = sum[1];
                        init_gmax = nan_if(sum[0] != sum[0] || sum[1] != sum[1], mx[0]);
                        trial_c = 1.0;
                        trial_phi = 0.5 * (sum[2] + a.cur.f_const);
                        trial_dphi = sum[3];
                        trial_gmax = nan_if(sum[2] != sum[2] || sum[3] != sum[3], mx[1]);
                        init_done = true;
                        z_zero = !ztrue_start && !Place::kResident;
                    }
                }
                if (!init_done)
                for_elems<T, EPT, US>(ld, tfirst, ps(), [&](int jj, int i) {
                    const bool valid = i < N;  // phantom slots run the generator but keep zeros
                    const NormalPair np = normal_pair(a.seed, sim, (uint64_t)i);
                    double zt, xt;
                    if constexpr (Model::kStencil) {
                        zt = sdk(jj, i) * np.n1;
                        xt = np.n2;                           // noise now, + A z after the barrier
                        g.set(jj, i, valid ? zt : 0.0);       // true z staged in the (still unused) gradient buffer
                    } else {
                        Model::sample(scoef(jj, i), np.n1, np.n2, zt, xt, i);
                    }
                    zt = valid ? zt : 0.0;
                    xt = valid ? xt : 0.0;
                    if constexpr (KEEP_ZTRUE) ztrue.set(jj, i, keep_value(valid, zt, sdk(jj, i), np, i));
                    x.set(jj, i, xt);
                    if (d.z0_mode == Z0_ZERO) z.set(jj, i, 0.0);
                    else if (d.z0_mode == Z0_TRUE) z.set(jj, i, zt);
                    else if constexpr (Place::kResident) z.set(jj, i, z0src.get(jj, i));
                }, when(Model::kStencil, g), when(KEEP_ZTRUE, ztrue), x, when(z_from_sample, z));
                if constexpr (!Place::kResident) {
                    if (!init_done && !z_from_sample && !z_in_place)
                        for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) { z.set(jj, i, z0src.get(jj, i)); }, z);
                }
            }
            if constexpr (Model::kStencil) {
                pass_barrier();
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const bool valid = i < N;
                    const int ic = valid ? i : 0;
                    const int im = ic == 0 ? (int)N - 1 : ic - 1, ip = ic == (int)N - 1 ? 0 : ic + 1;
                    const double az = fma(0.25, g.get1(im) + g.get1(ip), 0.5 * g.get1(ic));
                    const double xv = az + x.get(jj, i);
                    x.set(jj, i, valid ? xv : 0.0);
                }, x);
            }
        } else {
            VH xs;
            xs.bind(d.x_mode == X_DATA ? a.x_data : a.x_given, ld);
            x.clear(); g.clear(); z.clear(); s.clear();
            const bool z_zero = d.z0_mode == Z0_ZERO || d.z0_mode == Z0_TRUE;
            bool done = false;
            if constexpr (Place::kResident && Place::kXgLds) {
                if (pf_hit && pf.have_x && d.x_mode == X_DATA && !z_zero) {   // the data vector is in the x area already, the warm start here
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's LDS-DMA loads have landed
                    for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) { z.set(jj, i, z0src.get(jj, i)); });
                    done = true;
                } else if (pf_hit && pf.have_xg && d.x_mode == X_DATA && !z_zero) {   // ... in the g area: it travelled during the previous solve
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                        x.set(jj, i, g.get(jj, i));
                        z.set(jj, i, z0src.get(jj, i));
                    });
                    done = true;
                }
            }
            if (!done)
            for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                x.set(jj, i, xs.get(jj, i));
                if (z_zero) z.set(jj, i, 0.0);
                else if (!z_in_place) z.set(jj, i, z0src.get(jj, i));
            }, x, when(z_zero || !z_in_place, z));
        }
        if (pf_hit) {   // consumed -- or not taken from the x / g area (a data element that starts from zero: only the debug orders of
            // the deal reach that today): then its LDS-DMA loads must have landed before the solve writes gradients over them
            if constexpr (Place::kResident && Place::kXgLds) {
                if (pf.g_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            pf.p = -1;
            pf.g_pending = false;
        }
        stamp(p, 9);
        if constexpr (Model::kStencil) pass_barrier();  // x (and the start z) complete before neighbours read them
        else wg_barrier<true>();                        // (x of the LDS layout; elsewhere a thread reads back only its own elements)
        if constexpr (kBg && !KEEP_ZTRUE) {
            // arm the background generator with the cluster's next problem (same conditions as the foreground path above)
            gen.active = false;
            // (a launch that carries several maps: only a next problem of the SAME map -- the pairs are drawn and evaluated
            //  with the theta that is in LDS now; the first problem of another map is drawn in the foreground)
            const bool same_map = a.nmaps <= 1 || next_p / a.n_per_map == p / a.n_per_map;
            if (next_p >= 0 && d.tsample < 0 && same_map) {
                const ProblemDesc dn = describe(a, next_p);
                if (dn.x_mode == X_SAMPLE && dn.z0_mode == Z0_ZERO && !dn.normals_only && dn.tsample < 0) {
                    gen.p = next_p;
                    gen.trip = 0;
                    gen.sim = (uint64_t)dn.sim;
#pragma unroll
                    for (int k = 0; k < 4; ++k) gen.sum[k] = 0.0;
                    gen.mx[0] = gen.mx[1] = 0.0;
                    gen.active = true;
                    double* xb = bufsel ? wg_scratch : wg_scratch + (int64_t)(4 + 2 * kM + 1) * ld;
                    double* sb = bufsel ? wg_scratch + 2 * ld : xb + ld;
                    bg_xr = make_rsrc(xb, ld * 8);
                    bg_sr = make_rsrc(sb, ld * 8);
                }
            }
        }

        stamp(p, 1);
    }

    // The two-loop recursion of the streaming clusters with the NEXT step's first trips in flight across the cluster
    // exchange.  A step is a pass over two history vectors (A: the one the AXPY takes, B: the one the next coefficient's dot
    // product needs) that ends in a cluster-wide reduction (~3.5 us at 16 members: workgroup reduction, granules out, sweep);
    // the next step's coefficient needs that reduction -- its LOADS do not: which vectors it reads is known.  So before a
    // step's reduction the first trip of the next step's A and B is issued into the accessors' trip caches (BufChunk::preload)
    // and the pass finds them there.  The arithmetic and its order are untouched: the same bits.
    // (the stencil model's clusters: the elementwise models' solves are one iteration long, and their longer trips would cost
    //  32 registers of cache across the exchange)
    static constexpr bool kPrefetchTwoLoop = !Place::kResident && Place::kCluster && Model::kStencil;
    template <class F>
    __device__ __forceinline__ void pass_ab(VH& A, VH& B, F&& f) {   // the element loop of a two-loop step; writes s
        int t = tfirst;
        asm volatile("" : "+v"(t));
        const int n = (int)a.ld, pstr = pstride;
#pragma unroll 1
        for (int i0 = 2 * t; i0 < n; i0 += 2 * U * pstr) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 2 * u * pstr;
                f(2 * u, i);
                f(2 * u + 1, i + 1);
            }
            s.template flush<U>(i0, pstr);
            A.unload();   // (only the first trip was fetched ahead)
            B.unload();
        }
    }
    __device__ __forceinline__ double twoloop_prefetched(int upper, int lower, double dot0) {
        if constexpr (kPrefetchTwoLoop) {
            double dot = dot0;
            int t0 = tfirst;
            asm volatile("" : "+v"(t0));
            auto fetch = [&](VH& v) { v.template preload<U>(2 * t0, pstride); };
            VH A = hdg((upper - 1) % kM), B = A;   // (B: rebound where a step has one)
            if (upper > lower) B = hdx((upper - 2) % kM);
            // backward pass (q lives in s; the update pass left q = g there)
            for (int index = upper; index >= lower; --index) {
                const int slot = (index - 1) % kM;
                const double al = sh_rho[slot] * dot;
                if ((tid & 63) == 0) sh_alpha[slot] = al;  // lane 0 of EVERY wave: a wave reads back its own write (no barrier needed)
                double sum[1] = {0.0}, mx[1] = {0.0};
                if (index > lower) {
                    pass_ab(A, B, [&](int jj, int i) {
                        const double qi = fma(-al, A.get(jj, i), s.get(jj, i));
                        s.set(jj, i, qi);
                        sum[0] = fma(B.get(jj, i), qi, sum[0]);
                    });
                } else {  // last backward step: apply gamma = (dx.dg)/(dg.dg) of the newest pair
                    const double gam = sh_gam[(upper - 1) % kM];
                    pass_ab(A, B, [&](int jj, int i) {
                        const double dgi = A.get(jj, i);
                        const double si = gam * fma(-al, dgi, s.get(jj, i));
                        s.set(jj, i, si);
                        sum[0] = fma(dgi, si, sum[0]);
                    });
                }
                // the next step's vectors, their first trips on the way before this step's exchange
                if (index > lower) {
                    A = hdg((index - 2) % kM);
                    fetch(A);
                    if (index - 1 > lower) {
                        B = hdx((index - 3) % kM);
                        fetch(B);
                    }
                } else {   // the first forward step
                    A = hdx((lower - 1) % kM);
                    fetch(A);
                    if (lower < upper) {
                        B = hdg(lower % kM);
                        fetch(B);
                    } else {
                        fetch(g);
                    }
                }
                reduce<1, 0>(sum, mx);
                dot = sum[0];
            }
            // forward pass
            for (int index = lower; index <= upper; ++index) {
                const int slot = (index - 1) % kM;
                const double beta = sh_rho[slot] * dot;
                const double coef = sh_alpha[slot] - beta;
                double sum[1] = {0.0}, mx[1] = {0.0};
                if (index < upper) {
                    pass_ab(A, B, [&](int jj, int i) {
                        const double si = fma(A.get(jj, i), coef, s.get(jj, i));
                        s.set(jj, i, si);
                        sum[0] = fma(B.get(jj, i), si, sum[0]);
                    });
                    A = hdx(index % kM);
                    fetch(A);
                    if (index + 1 < upper) {
                        B = hdg((index + 1) % kM);
                        fetch(B);
                    } else {
                        fetch(g);
                    }
                } else {
                    pass_ab(A, g, [&](int jj, int i) {
                        const double si = -fma(A.get(jj, i), coef, s.get(jj, i));
                        s.set(jj, i, si);
                        sum[0] = fma(g.get(jj, i), si, sum[0]);
                    });
                }
                reduce<1, 0>(sum, mx);
                dot = sum[0];
            }
            return dot;
        } else {
            (void)upper; (void)lower;
            return dot0;
        }
    }

    // -- phase 2: zhat_at_theta -- Optim LBFGS + HagerZhang on -logLike from the z prepared by begin()
    __device__ __forceinline__ void solve(int p) {
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
        bool init_fused = kFuseInit;  // s = -g and g . s are in hand when the first iteration starts
        if (init_done) {              // begin() evaluated the initial point (and the first trial) while sampling
            f = init_f;
            dphi_init = init_dphi;
            gmax = init_gmax;
            f_calls = 1;
            have_trial = true;
            init_fused = true;
        } else if constexpr (kFuseInit) {
            eval_init_with_trial(1.0, f, dphi_init, gmax);
        } else {
            have_trial = false;
            eval<false, true, Model::kStencil>(0.0, f, dphi_init, gmax);  // stencil: g stored, s = -g with it
            if constexpr (Model::kStencil) init_fused = true;
        }
        bool g_stored = !init_fused || Model::kStencil;
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
            if (init_fused && iterations == 1) {
                dphi_0 = dphi_init;  // s = -g and g . s came with the initial evaluation
            } else if (h == 0 || !have_pair) {
                double sum[1] = {0.0}, mx[1] = {0.0};
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const double gi = g.get(jj, i), si = -gi;
                    s.set(jj, i, si);
                    sum[0] = fma(gi, si, sum[0]);
                }, s);
                reduce<1, 0>(sum, mx);
                dphi_0 = sum[0];
            } else if constexpr (kPrefetchTwoLoop) {
                hist_words += h;
                dphi_0 = twoloop_prefetched(upper, lower, dot0);
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
                        for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                            const double qi = fma(-al, dgp.get(jj, i), s.get(jj, i));
                            s.set(jj, i, qi);
                            sum[0] = fma(dxn.get(jj, i), qi, sum[0]);
                        }, s);
                    } else {  // last backward step: apply gamma = (dx.dg)/(dg.dg) of the newest pair
                        const double gam = sh_gam[(upper - 1) % kM];
                        for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
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
                        for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                            const double si = fma(dxp.get(jj, i), coef, s.get(jj, i));
                            s.set(jj, i, si);
                            sum[0] = fma(dgn.get(jj, i), si, sum[0]);
                        }, s);
                    } else {
                        for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
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
                if constexpr (!Model::kStencil) {
                    drop_g_prefetch();
                    if (!g_stored) {  // (cannot happen after a finite, unconverged initial evaluation; kept for completeness)
                        for_elems_zz([&](auto zz, int jj, int i) {
                            double unused = 0.0;
                            const double zi = decltype(zz)::value ? 0.0 : z.get(jj, i);
                            g.set(jj, i, Model::grad(gcoef(jj, i), x.get(jj, i), zi, unused, i));
                        }, g);
                        g_stored = true;
                    }
                }
                double sum[1] = {0.0}, mx[1] = {0.0};
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
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
            spec_c = NAN;
            spec_on = iterations == 1 && !(a.debug & 32);   // (MUSE_DEBUG bit 5: never speculate -- A/B and tests)
            double alpha;
            iter_stamp = iterations == 1 ? 0 : 99;
            stamp_p = p;
            const bool ls_ok = linesearch(1.0, phi_0, dphi_0, alpha);
            have_trial = false;  // only the first line search can use the trial evaluated with the initial point
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
                double acc[KB];  // (big tier: ONE sum over all blocks -- the NaN channel below; finish() forms the block sums)
#pragma unroll
                for (int b = 0; b < KB; ++b) acc[b] = 0.0;
                VH zout;
                const bool store = Place::kResident && d.zslot >= 0;  // streaming: z already lives in its zhat slot
                if (store) zout.bind(a.zhat + d.zslot * ld, ld);
                // the accepted step is a trial that speculated (eval, SPEC): its pass did all of this already
                bool spec_hit = false;
                if constexpr (kSpec) spec_hit = alpha == spec_c;   // (NaN: no such trial)
                if (spec_hit) {
                    if constexpr (kSpec) {
#pragma unroll
                        for (int b = 0; b < KB; ++b) acc[b] = spec_acc[b];
                        mx[0] = spec_mx;
                        if constexpr (Place::kResident) {   // what is left of the pass: the step itself and the MAP out -- no score terms, no reduction
                            for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                                const double zn = fma(alpha, s.get(jj, i), z.get(jj, i));
                                z.set(jj, i, zn);
                                if (store) zout.set(jj, i, zn);
                            });
                        }
                    }
                } else {
                for_elems_zz_r(reads(s), reads(x), [&](auto zz, int jj, int i) {
                    const double zo = decltype(zz)::value ? 0.0 : z.get(jj, i), si = s.get(jj, i);
                    const double zn = fma(alpha, si, zo);
                    z.set(jj, i, zn);
                    mx[0] = absmax(mx[0], zn - zo);
                    if constexpr (Place::kResident) {
                        if (store) zout.set(jj, i, zn);
                    }
                    score_add<KB>(acc, x.get(jj, i), zn, jj, i);
                }, z);
                if (iterations == 1) stamp(p, 14);
                if constexpr (KB + 1 <= 8) {
                    reduce<KB, 1>(acc, mx);
                    if (iterations == 1) stamp(p, 15);
                } else {
                    double none[1] = {0.0};
                    reduce<0, 1>(none, mx);
                    reduce<KB, 0>(acc, none);
                }
                }
                bool any_nan = false;  // a NaN step shows up in the score sums (NaN channel of the maximum)
#pragma unroll
                for (int b = 0; b < KB; ++b) {
                    score_acc[b] = acc[b];
                    any_nan = any_nan || acc[b] != acc[b];
                }
                mx[0] = nan_if(any_nan, mx[0]);
                score_ready = !kBig;
            } else if constexpr (!Model::kStencil) {
                auto body = [&](auto have_g, auto zz, int jj, int i) {
                    const double zo = decltype(zz)::value ? 0.0 : z.get(jj, i), si = s.get(jj, i);
                    const double dxi = alpha * si;
                    const double zn = fma(alpha, si, zo);  // the same point the accepted trial evaluated
                    z.set(jj, i, zn);
                    mx[0] = absmax(mx[0], zn - zo);
                    double unused = 0.0;
                    const double xi = x.get(jj, i);
                    const auto ivi = gcoef(jj, i);
                    const double gn = Model::grad(ivi, xi, zn, unused, i);
                    double go;
                    if constexpr (decltype(have_g)::value) go = g.get(jj, i);
                    else go = Model::grad(ivi, xi, zo, unused, i);  // what the initial evaluation computed
                    const double dgi = gn - go;
                    sum[0] = fma(dxi, dgi, sum[0]);
                    sum[1] = fma(dgi, dgi, sum[1]);
                    sum[2] = fma(dxi, gn, sum[2]);
                    dxs.set(jj, i, dxi);
                    dgs.set(jj, i, dgi);
                    g.set(jj, i, gn);
                    s.set(jj, i, gn);
                };
                drop_g_prefetch();
                if (g_stored) {
                    for_elems_zz([&](auto zz, int jj, int i) { body(std::true_type{}, zz, jj, i); }, z, dxs, dgs, g, s);
                } else {
                    for_elems_zz([&](auto zz, int jj, int i) { body(std::false_type{}, zz, jj, i); }, z, dxs, dgs, g, s);
                    g_stored = true;
                }
                reduce<3, 1>(sum, mx);
                mx[0] = nan_if(sum[0] != sum[0], mx[0]);
            } else {
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
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
            z_zero = false;  // z has been written
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
        if constexpr (!Place::kResident && !Model::kStencil) {
            if (z_zero) {  // no step was taken from a zero start (converged at once / non-finite): z = 0 goes to memory now
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) { z.set(jj, i, 0.0); }, z);
                z_zero = false;
            }
        }

        stamp(p, 6);
    }

    // What the implicit-differentiation H keeps of the draw in its extra vector: the simulation's true z (compiled-in models: x's
    // derivative in theta follows from it) or, for a user-supplied model, dx_i / dtheta_k itself (include/muse_model.h).
    __device__ __forceinline__ double keep_value(bool valid, double zt, double sd, const NormalPair& np, int i) const {
        if constexpr (Model::kId == MUSE_MODEL_USER) return valid ? Model::dx_dtheta(sd, np.n1, np.n2, i) : 0.0;
        else return zt;
    }

    // ------------------------------------------------------------------------------------------
    // get_H! implicit-differentiation branch for one simulation (src/muse.jl:335-405):
    //   H = H1 - dFdtheta^T A^{-1} dFdtheta1,  A = Hessian_z logLike at (x, zhat, theta0),
    // A^{-1} by conjugate gradients (IterativeSolvers.cg: x0 = 0, reltol sqrt(eps), abstol 0, maxiter).
    // The reference gets the derivative operands by nested AD; for the compiled-in models they are
    // closed forms (see oracle/muse_oracle.c, mo_implicit_H, for the list).  Streaming policy only:
    // the CG vectors reuse the solver's g, s and history buffers; z_true sits in the extra vector.
    // Writes H[p] (row-major ntheta x ntheta) and the CG iteration count of column j to info[p*ntheta+j].
    __device__ __forceinline__ void run_implicit(int p, double* wg_scratch, double* lds_x, double* lds_g) {
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
        // imp_split == ntheta: this element is ONE column of one simulation's H (few simulations, many theta: the
        // columns spread over the GPU, each repeating the cheap atol = 1e-1 MAP); imp_split == 1: all columns
        const int split = a.imp_split > 1 ? a.imp_split : 1;
        const int plist = p + a.p0;  // position in the list that starts at sim_begin's first column
        const int64_t psim = plist / split;
        const int j_lo = split > 1 ? plist % split : 0, j_hi = split > 1 ? j_lo + 1 : nth;
        for (int j = j_lo; j < j_hi; ++j) {
            // ---- right-hand side b = dFdtheta1[:, j]; v = 0, r = p = b --------------------------------
            double sum[1] = {0.0}, mx[1] = {0.0};
            if constexpr (Model::kStencil) {
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const double zt = ztrue.get(jj, i);  // unconditional: the pair load is issued at the even element
                    t1.set(jj, i, blk(jj, i) == j ? 0.5 * zt : 0.0);
                }, t1);
                pass_barrier();
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) { t2.set(jj, i, Aat(t1, i)); }, t2);
                pass_barrier();
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const double bi = Aat(t2, i);
                    v.set(jj, i, 0.0);
                    r.set(jj, i, bi);
                    pp.set(jj, i, bi);
                    sum[0] = fma(bi, bi, sum[0]);
                }, v, r, pp);
            } else {
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const double zt = ztrue.get(jj, i);  // unconditional: the pair load is issued at the even element
                    double bi;
                    if constexpr (Model::kId == MUSE_MODEL_NOISE) bi = iv0 * (0.5 * (x.get(jj, i) - zt));
                    else if constexpr (Model::kId == MUSE_MODEL_USER) {
                        // zt is dx_i / dtheta_k here (keep_value); the element's d2 o / dz2 goes to t1 for the CG passes
                        double ozz, ozx, bz, bx;
                        Model::second(ivk(jj, i), x.get(jj, i), z.get(jj, i), ozz, ozx, bz, bx, i);
                        t1.set(jj, i, ozz);
                        const double bb = -(ozx * zt);
                        bi = blk(jj, i) == j ? bb : 0.0;
                    }
                    else bi = blk(jj, i) == j ? 0.5 * zt : 0.0;
                    v.set(jj, i, 0.0);
                    r.set(jj, i, bi);
                    pp.set(jj, i, bi);
                    sum[0] = fma(bi, bi, sum[0]);
                }, v, r, pp, when(Model::kId == MUSE_MODEL_USER, t1));
            }
            reduce<1, 0>(sum, mx);
            double rr = sum[0];
            const double tol = __builtin_sqrt(kEps) * __builtin_sqrt(rr);
            int it = 0;
            while (it < a.cg_maxiter && !(__builtin_sqrt(rr) <= tol)) {
                // ---- Ap = A_hess p, p.Ap -----------------------------------------------------------
                double s1[1] = {0.0};
                if constexpr (Model::kStencil) {
                    for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) { t1.set(jj, i, Aat(pp, i)); }, t1);
                    pass_barrier();
                    for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                        const double pi = pp.get(jj, i);
                        const double api = -(Aat(t1, i) + ivk(jj, i) * pi);
                        Ap.set(jj, i, api);
                        s1[0] = fma(pi, api, s1[0]);
                    }, Ap);
                } else {
                    for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                        const double pi = pp.get(jj, i);
                        double api;
                        if constexpr (Model::kId == MUSE_MODEL_NOISE) api = -((iv0 + 1.0) * pi);
                        else if constexpr (Model::kId == MUSE_MODEL_USER) api = -(t1.get(jj, i) * pi);
                        else api = -(pi + ivk(jj, i) * pi);
                        Ap.set(jj, i, api);
                        s1[0] = fma(pi, api, s1[0]);
                    }, Ap);
                }
                reduce<1, 0>(s1, mx);
                const double alpha = rr / s1[0];
                // ---- v += alpha p ; r -= alpha Ap ; r.r ---------------------------------------------
                double s2[1] = {0.0};
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    v.set(jj, i, fma(alpha, pp.get(jj, i), v.get(jj, i)));
                    const double ri = fma(-alpha, Ap.get(jj, i), r.get(jj, i));
                    r.set(jj, i, ri);
                    s2[0] = fma(ri, ri, s2[0]);
                }, v, r);
                reduce<1, 0>(s2, mx);
                const double beta = s2[0] / rr;
                rr = s2[0];
                // ---- p = r + beta p (its stores are ordered before the next stencil read by pass_barrier)
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    pp.set(jj, i, fma(beta, pp.get(jj, i), r.get(jj, i)));
                }, pp);
                if constexpr (Model::kStencil) pass_barrier();
                it += 1;
            }
            // ---- H[:, j] = H1[:, j] - dFdtheta^T v ----------------------------------------------------
            // (big tier: eight rows of the column per pass, as finish() forms its block sums)
            constexpr int NB = kBig ? 8 : MAXB;
            constexpr int KA = Model::kId == MUSE_MODEL_NOISE ? 2 : NB;  // noise: [dFdtheta^T v, H1 sum]
            double h1u[1] = {0.0};  // user model: sum_{i in block j} bx_i dx_i/dtheta_j
#pragma unroll 1
            for (int c = 0; c < (kBig ? nth : 1); c += 8) {
                double acc[KA];
#pragma unroll
                for (int b = 0; b < KA; ++b) acc[b] = 0.0;
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const double zi = z.get(jj, i), vi = v.get(jj, i);
                    if constexpr (Model::kId == MUSE_MODEL_NOISE) {
                        const double xi = x.get(jj, i);
                        const double dd = xi - zi;
                        acc[0] = fma(-iv0 * dd, vi, acc[0]);
                        acc[1] = fma(dd, 0.5 * (xi - ztrue.get(jj, i)), acc[1]);
                    } else if constexpr (Model::kId == MUSE_MODEL_USER) {
                        const double ivi = ivk(jj, i), tx = ztrue.get(jj, i);
                        double ozz, ozx, bz, bx;
                        Model::second(ivi, x.get(jj, i), zi, ozz, ozx, bz, bx, i);
                        const double t = 0.5 * (ivi * bz);
                        const int kf = blk(jj, i), k = kf - c;
#pragma unroll
                        for (int b = 0; b < NB; ++b) acc[b] = (k == b) ? fma(t, vi, acc[b]) : acc[b];
                        h1u[0] = (kf == j && c == 0) ? fma(bx, tx, h1u[0]) : h1u[0];
                    } else {
                        const double t = ivk(jj, i) * zi;
                        if constexpr (MAXB == 1) {
                            acc[0] = fma(t, vi, acc[0]);
                        } else {
                            const int k = blk(jj, i) - c;
#pragma unroll
                            for (int b = 0; b < NB; ++b) acc[b] = (k == b) ? fma(t, vi, acc[b]) : acc[b];
                        }
                    }
                });
                reduce<KA, 0>(acc, mx);
                if constexpr (Model::kId == MUSE_MODEL_USER) {
                    if (c == 0) reduce<1, 0>(h1u, mx);
                }
                if (tid == 0 && crank == 0) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (c + b < nth) {
                            double h1 = 0.0;
                            if constexpr (Model::kId == MUSE_MODEL_NOISE) h1 = iv0 * acc[1];
                            if constexpr (Model::kId == MUSE_MODEL_USER) {
                                const double ivj = kBig ? big_iv[j] : a.cur.t.iv[j];
                                h1 = c + b == j ? 0.5 * (ivj * h1u[0]) : 0.0;
                            }
                            a.scores[(psim * nth + c + b) * nth + j] = h1 - acc[b];
                        }
                    }
                }
            }
            if (tid == 0 && crank == 0) {
                muse_info inf;
                inf.iterations = it;
                inf.f_calls = f_calls;
                inf.status = status;
                inf.hist_words = hist_words;
                inf.f_min = f;
                inf.gnorm = gmax;
                a.info[psim * nth + j] = inf;
            }
        }
    }

    // -- phase 3: zhat out, score grad_theta logLike(x, zhat, theta), solver info
    __device__ __forceinline__ void finish(int p) {
        const int64_t ld = a.ld;
        if constexpr (kBig) {
            // the big tier: the block sums of the score terms eight blocks at a time -- a pass over z (and x) and one
            // reduction per chunk, the summation tree of a block being that of the small tiers
            const int nth = a.ntheta;
#pragma unroll 1
            for (int c = 0; c < nth; c += 8) {
                double acc[8], mx[1] = {0.0};
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[b] = 0.0;
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const double t = Model::score_term(x.get(jj, i), z.get(jj, i), i);
                    const int k = blk(jj, i) - c;
#pragma unroll
                    for (int b = 0; b < 8; ++b) acc[b] += (k == b) ? t : 0.0;
                });
                reduce<8, 0>(acc, mx);
                int tl = tid;
                asm volatile("" : "+v"(tl));
                const int kk = c + tl;
                if (tl < 8 && kk < nth && crank == 0) {
                    double mine = acc[0];
#pragma unroll
                    for (int b = 1; b < 8; ++b) mine = (tl == b) ? acc[b] : mine;
                    const long long lo = ((long long)kk * a.N + nth - 1) / nth, hi = ((long long)(kk + 1) * a.N + nth - 1) / nth;
                    a.scores[d.row * nth + kk] = 0.5 * (big_iv[kk] * mine - (double)(hi - lo));
                }
            }
            if (tid == 0 && crank == 0) {
                muse_info inf;
                inf.iterations = iterations;
                inf.f_calls = f_calls;
                inf.status = (status <= MUSE_STATUS_F_CONVERGED && !isfinite(f)) ? MUSE_STATUS_NONFINITE : status;
                inf.hist_words = hist_words;
                inf.f_min = f;
                inf.gnorm = gmax;
                a.info[info_row(p)] = inf;
            }
        } else {
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
                for_elems<T, EPT, U>(ld, tfirst, ps(), [&](int jj, int i) {
                    const double zi = z.get(jj, i);
                    if constexpr (Place::kResident) {
                        if (store) zout.set(jj, i, zi);
                    }
                    score_add<MAXB>(acc, x.get(jj, i), zi, jj, i);
                });
                reduce<MAXB, 0>(acc, mx);
            }
            int tl = tid;
            asm volatile("" : "+v"(tl));
            if (tl < MAXB && tl < a.ntheta && crank == 0) {  // lane b finishes and writes score component b
                double mine = acc[0];
#pragma unroll
                for (int b = 1; b < MAXB; ++b) mine = (tl == b) ? acc[b] : mine;
                double sc;
                if constexpr (Model::kPair) {
                    // lane tl writes component tl of the score: parameter tl / K of block k = tl mod K, from the block's two sums
                    // (acc[k], acc[K + k]), its coefficients and its element count -- assembled by the header (muse_model_score)
                    const int K = a.ntheta >> 1, k = tl < K ? tl : tl - K;
                    double s0 = acc[0], s1 = acc[0];
#pragma unroll
                    for (int b = 1; b < MAXB; ++b) {
                        s0 = (k == b) ? acc[b] : s0;
                        s1 = (k + K == b) ? acc[b] : s1;
                    }
                    const double cnt = (double)((k == K - 1 ? (int)a.N : a.bnd32[k + 1]) - a.bnd32[k]);
                    double ga, gb;
                    Model::score(pair_table(a.cur.t) + 4 * k, s0, s1, cnt, ga, gb);
                    sc = tl < K ? ga : gb;
                    (void)mine;
                } else {
                const double cnt = (double)(a.bnd32[tl < a.ntheta - 1 ? tl + 1 : 0] - a.bnd32[tl]);
                const double cnt_last = (double)((int)a.N - a.bnd32[tl]);  // bnd32[ntheta] is a sentinel, not N
                sc = 0.5 * (a.cur.t.iv[tl] * mine - (tl == a.ntheta - 1 ? cnt_last : cnt));
                }
                a.scores[d.row * a.ntheta + tl] = sc;
                if (a.gran) {  // device-resident muse! loop: the component also leaves as two tagged granules (args.hpp)
                    typedef __attribute__((address_space(1))) unsigned long long gu64;
                    gu64* gq = (gu64*)a.gran + (d.row * a.ntheta + tl) * 2;
                    const unsigned long long b = (unsigned long long)__double_as_longlong(sc), tg = (unsigned long long)a.gran_tag << 32;
                    if (a.gran_sys == 2) {   // a board per GPU, in device memory: the granules go into every rank's (kernarg: gran_peers)
                        const BatchArgs* ka = reinterpret_cast<const BatchArgs*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
                        const int64_t off = (int64_t)((gu64*)gq - (gu64*)a.gran);
                        for (int q = 0; q < ka->ngran_peers; ++q) {
                            gu64* pq = (gu64*)ka->gran_peers[q] + off;
                            __hip_atomic_store(pq, tg | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(pq + 1, tg | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    } else if (a.gran_sys) {   // the node's board in host memory: other GPUs' steppers read it
                        __hip_atomic_store(gq, tg | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(gq + 1, tg | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    } else {
                        __hip_atomic_store(gq, tg | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(gq + 1, tg | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (tid == 0 && crank == 0) {
                muse_info inf;
                inf.iterations = iterations;
                inf.f_calls = f_calls;
                inf.status = (status <= MUSE_STATUS_F_CONVERGED && !isfinite(f)) ? MUSE_STATUS_NONFINITE : status;
                inf.hist_words = hist_words;
                inf.f_min = f;
                inf.gnorm = gmax;
                a.info[info_row(p)] = inf;
            }
        }
        stamp(p, 7);
    }
    double last_phi;
    double score_acc[KB];  // per-block sums of the score terms when the solve's last pass computed them
    bool score_ready;
};


}  // namespace muse
