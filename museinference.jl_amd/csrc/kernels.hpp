// kernels.hpp -- the solver kernel, the loop kernel and their launch shims as templates: every device translation unit of the engine
// includes this; muse_kernels.hip holds the dispatch and the per-simulation operator kernels, kernels_part.hip -- compiled once per
// group of models (-DMUSE_PART=n, build.py), side by side -- the instantiations (one unit took 6.5 minutes; eight take ~1.5).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/muse_hip.h"
#include "solver.hpp"
#include "step.hpp"

namespace muse {



// The argument block is read from an LDS copy of the kernarg segment, not from the by-value
// parameter: hipcc materialises a by-value aggregate in scratch as soon as any select/phi of two
// field addresses is formed, and every access then becomes a scratch access.  LDS loads at uniform
// addresses are uniform values, so control flow on them stays scalar.
typedef __attribute__((address_space(4))) const uint32_t* kernarg_ptr;
// theta of problem p into the LDS copy of the arguments (BatchArgs::cur) -- a launch that carries several maps -- from the
// map's entry of maps[], read straight from the kernarg segment.  Workgroup-uniform; a no-op (no barrier) for the plain launch.
template <bool RAW>
__device__ __forceinline__ void load_problem_theta(const BatchArgs& a, double* args_lds, int p, int tid) {
    if (a.nmaps > 1) {
        wg_barrier<RAW>();  // every thread is done with the previous problem's theta
        asm volatile("" : "+v"(tid));  // (else the source address is formed at the kernel's entry and held -- spilled -- across it)
        if (tid < (int)(sizeof(MapTheta) / 4)) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(args_lds) + offsetof(BatchArgs, cur) / 4;
            kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            dst[tid] = kp[(offsetof(BatchArgs, maps) + (size_t)(p / a.n_per_map) * sizeof(MapTheta)) / 4 + tid];
        }
        wg_barrier<RAW>();
    }
}

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
#ifdef MUSE_STAMPS
    unsigned long long t_entry;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry)::"memory");
#endif
    {
        kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t* dst = reinterpret_cast<uint32_t*>(args_lds);
        for (int w = tid; w < (int)(kArgsHeadBytes / 4); w += T) dst[w] = kp[w];  // (not the trailing maps[]: see load_problem_theta)
        if (tid == 0) ticket[2] = 0;   // (solver.hpp, wait_fiducial: the tag of the fiducial MAP this workgroup has seen published)
    }
    __syncthreads();
    const BatchArgs& a = *reinterpret_cast<const BatchArgs*>(args_lds);
    double* exch = args_lds + kArgsDoubles;   // cluster placements: [kMaxCluster][8] values of the epoch's exchange
    double* lds_x = exch + (Place::kCluster ? kMaxCluster * 8 : 0);  // [ld + 2]: elements, dummy slot (index ld), pad
    double* lds_g = lds_x + a.ld + 2;         // [ld + 2]
    // (only when asked for: no stamp executes in a timed launch.  Every lane of wave 0 stores the same scalars: the condition
    // lives in a scalar register, where a per-lane `tid == 0` would keep a vector register alive to the kernel's last line)
    const bool wave0 = (__builtin_amdgcn_readfirstlane(tid) >> 6) == 0;
    auto clock_stamp = [&](int k) {
        if (a.clock_out && blockIdx.x == 0 && wave0) {
            unsigned long long t0, t1;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(t1)::"memory");
            a.clock_out[2 * k] = t0;
            a.clock_out[2 * k + 1] = t1;
        }
    };
    clock_stamp(0);
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
        // XCD-local clusters: workgroup b runs on XCD b mod 8 (tools/xcdprobe.hip: 0 of 512 workgroups elsewhere), so a
        // cluster takes its members from one residue class (cluster = (w / csize) * 8 + x, rank = w mod csize for
        // b = 8 w + x): an agent-scope flag hand-off inside an XCD costs ~1030 shader cycles against ~1510 across XCDs
        // (workgroup-scope accesses do not see another CU's stores at all).  Only a matter of speed: the accesses are
        // agent-scope either way.
        const int csize = a.csize;
        const int cluster = a.xcd_local ? ((int)(blockIdx.x >> 3) / csize) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x / csize;
        const int crank = a.xcd_local ? (int)(blockIdx.x >> 3) % csize : (int)blockIdx.x % csize;
        double* cl_scratch = a.scratch + (int64_t)cluster * a.scratch_stride;
        Solver<Model, Place> sv(a, tid, red, shs);
        sv.crank = crank;
        sv.csize = csize;
        sv.tfirst = crank * T + tid;
        sv.pstride = csize * T;
        sv.pack_blocks();
        sv.cl_part = a.cl_part + (size_t)cluster * kClusterSlotDoubles;
        sv.exch = exch;
        sv.cl_epoch = a.cl_state[cluster];  // granule tags continue across launches
        int nth = 0;
#ifdef MUSE_STAMPS
        if (tid == 0 && a.stamps && cluster < a.nproblems) a.stamps[(size_t)cluster * 16 + 8] = t_entry;  // kernel entry
#endif
        for (int p = cluster; p < a.nproblems; p += a.nclusters, ++nth) {
            sv.bufsel = nth & 1;
            sv.next_p = p + a.nclusters < a.nproblems ? p + a.nclusters : -1;
            sv.parity = 0;
            wg_barrier<!Model::kStencil>();
            load_problem_theta<!Model::kStencil>(a, args_lds, p, tid);
            if constexpr (IMPLICIT) sv.run_implicit(p, cl_scratch, lds_x, lds_g);
            else sv.run(p, cl_scratch, lds_x, lds_g);
        }
#ifdef MUSE_STAMPS
        if (cluster < a.nproblems) sv.stamp(cluster + ((a.nproblems - 1 - cluster) / a.nclusters) * a.nclusters, 9);  // last instruction but the epoch store
#endif
        clock_stamp(1);
        if (tid == 0 && crank == 0) {
            a.cl_state[cluster] = sv.cl_epoch;
            if (cluster == 0) a.error_flag[1] = (int)(sv.cl_epoch >> 1);  // the host resets the tags long before a wrap
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
        // Problems are dealt dynamically, but no workgroup waits for the dealer: the first problem is the workgroup's own
        // index (the host launches grid <= nproblems), and the ticket of the NEXT problem (p = grid + ticket) is drawn when
        // the current one begins -- the atomic's round trip (~1.5 k cycles, three of them per workgroup at configs[1]) is
        // over long before its result is looked at.
        int p = (int)blockIdx.x;
        for (;;) {
            int next = 0;
            if (tid == 0) next = atomicAdd(a.work_counter, 1);
#ifdef MUSE_STAMPS
            if (tid == 0 && a.stamps && p < (int)gridDim.x) a.stamps[(size_t)p * 16 + 8] = t_entry;  // kernel entry
#endif
            load_problem_theta<!Model::kStencil>(a, args_lds, p, tid);
            {
                Solver<Model, Place> sv(a, tid, red, shs);
                sv.pk[0] = pk0;
                sv.pk[1] = pk1;
                if constexpr (IMPLICIT) sv.run_implicit(p, wg_scratch, lds_x, lds_g);
                else sv.run(p, wg_scratch, lds_x, lds_g);
            }
            // (raw barriers for the elementwise models: the MAP's stores keep draining while the next problem starts)
            wg_barrier<!Model::kStencil>();
            if (tid == 0) ticket[0] = next;
            wg_barrier<!Model::kStencil>();
            p = (int)gridDim.x + (__builtin_amdgcn_readfirstlane(ticket[0]) - a.ticket_base);
            if (p >= a.nproblems) break;
        }
        clock_stamp(1);
    }
}

// ------------------------------------------------------------------------------------------------
// The device-resident muse! loop (muse_run_device, muse_engine.cpp): ONE launch runs every outer iteration
// (src/muse.jl:159-232).  The grid is sized so that every workgroup is resident (loop_max_grid); workgroup w owns elements
// w, w + grid, ... in EVERY iteration (an element's MAP stays with the compute unit that wrote it).  An iteration:
//   1. the map: the workgroup's elements, exactly as map_score_kernel runs them (Solver::run); every element also
//      publishes its score as tagged granules (Solver::finish);
//   2. the exchange: every workgroup sweeps the (nsims + 1) x ntheta x 2 granules until all carry this iteration's tag --
//      no fence, no barrier, no atomic counter (the argument is solver.hpp's cluster_exchange: an aligned 8-byte granule
//      written by one write-through store and read past the L1 is not torn and needs no ordering against a flag);
//   3. the step, by EVERY workgroup for itself from the same bits (step.hpp: moments in the fixed 64-leaf tree, one wavefront
//      per theta component; the dense part on one lane, its arrays in LDS): the history record, the next theta -- written
//      into the workgroup's LDS copy of the arguments, where its next problems read it -- and the convergence test
//      (src/muse.jl:163-166).  Workgroup 0 also writes the record, the iterate and the status to pinned host memory.
// Nothing leaves the GPU between two iterations and no launch sits between them.  The arithmetic is the host loop's
// (muse_run), so the two trajectories are the same bits.
constexpr int kLoopArgsDoubles = (int)((sizeof(LoopArgs) + 7) / 8);
constexpr int kLoopRecDoubles = 7 * kMaxTheta + kMaxTheta * kMaxTheta + 1;
constexpr int kLoopPersistDoubles = kLoopArgsDoubles + kLoopRecDoubles + 3;   // LoopArgs, the previous record, flags
enum { STEP_TIMEOUT = 100 };  // a workgroup's sweep of the granules expired (the workgroups were not all resident)

// Where things are in the loop kernel's LDS, re-derived where they are used from a laundered zero: computed once at the
// kernel's entry, the dozen addresses and the loop's scalars stay live across the problems of every iteration and spill
// (13 VGPRs and 95 SGPRs in the first version).
template <class Place>
struct LoopLds {
    BatchArgs* a;
    double *red, *shs, *lds_x, *lds_g, *persist, *stepbuf, *rec;
    LoopArgs* L;
    int* flags;                 // [0] err, [1] converged, [2] sweep expired
    unsigned long long* t_prev; // s_memrealtime at the end of the previous step (workgroup 0's is the one that is reported)
    __device__ __forceinline__ LoopLds(unsigned char* smem) {
        unsigned z = 0;
        asm volatile("" : "+s"(z));
        red = reinterpret_cast<double*>(smem + z);
        shs = red + 2 * (Place::T / 64) * 8;
        double* args_lds = shs + 42;
        a = reinterpret_cast<BatchArgs*>(args_lds);
        lds_x = args_lds + kArgsDoubles;
        lds_g = lds_x + a->ld + 2;
        // after x and g (or right behind the arguments): what lives across iterations; the step's own arrays alias x and g
        // in the LDS-resident layout (both are dead between two iterations), and follow the persistent block otherwise
        persist = Place::kXgLds ? lds_g + a->ld + 2 : lds_x;
        stepbuf = Place::kXgLds ? lds_x : persist + kLoopPersistDoubles;
        L = reinterpret_cast<LoopArgs*>(persist);
        rec = persist + kLoopArgsDoubles;
        flags = reinterpret_cast<int*>(rec + kLoopRecDoubles);
        t_prev = reinterpret_cast<unsigned long long*>(rec + kLoopRecDoubles + 2);
    }
};

// The per-block tables of a new theta for the two-parameter family (include/muse_model.h, MUSE_MODEL_PAIR): lane k < K evaluates the
// HEADER's muse_model_coefs for block k -- the statements the host runs in pair_map_theta (muse_engine.cpp), so the same bits -- and
// lane 0 the constant term in block order.
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
__device__ __forceinline__ void pair_theta_update(int lane, int nt, const int64_t* bnd, const double* th, MapTheta& m) {
    const int K = nt >> 1;
    if (lane < K) {
        double cf[4] = {0.0, 0.0, 0.0, 0.0};
        (void)muse_model_coefs(th[lane], th[K + lane], cf);
        m.t.theta[lane] = th[lane];
        m.t.theta[K + lane] = th[K + lane];
        double* rec = &m.t.sd[0] + 4 * lane;   // (models.hpp, pair_table: sd and iv as ONE [block][4] table)
        rec[0] = cf[0];
        rec[1] = cf[1];
        rec[2] = cf[2];
        rec[3] = cf[3];
    } else if (lane < kMaxTheta / 2) {   // (the records beyond, and their parameters: as pair_map_theta's memset leaves them)
        double* rec = &m.t.sd[0] + 4 * lane;
        rec[0] = rec[1] = rec[2] = rec[3] = 0.0;
    }
    if (lane >= 2 * K && lane < kMaxTheta) m.t.theta[lane] = 0.0;
    if (lane == kMaxTheta) {
        double cst = 0.0;
        for (int k = 0; k < K; ++k) {
            double cf[4] = {0.0, 0.0, 0.0, 0.0};
            const double C = muse_model_coefs(th[k], th[K + k], cf);
            cst += (double)(bnd[k + 1] - bnd[k]) * C;
        }
        m.f_const = cst;
        m.pad_ = 0.0;
    }
}
#endif

template <class Model, class Place>
__global__ void __launch_bounds__(Place::T) __attribute__((amdgpu_waves_per_eu(Place::kWavesPerEu)))
muse_loop_kernel(const BatchArgs /*read via the kernarg segment*/, const LoopArgs /*likewise*/) {
    static_assert(!Place::kCluster, "one workgroup per element");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = Place::T, NW = T / 64;
    const int tid = threadIdx.x;
    {
        kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t* dst = reinterpret_cast<uint32_t*>(reinterpret_cast<double*>(smem) + 2 * (T / 64) * 8 + 42);
        for (int w = tid; w < (int)(kArgsHeadBytes / 4); w += T) dst[w] = kp[w];
    }
    __syncthreads();
    {
        LoopLds<Place> m(smem);
        kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t* dst = reinterpret_cast<uint32_t*>(m.persist);
        for (int w = tid; w < (int)(sizeof(LoopArgs) / 4); w += T) dst[w] = kp[sizeof(BatchArgs) / 4 + w];
        if (tid == 0) {
            unsigned long long now;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            *m.t_prev = now;
            m.flags[0] = m.flags[1] = m.flags[2] = 0;
        }
    }
    unsigned pk0, pk1;
    {
        LoopLds<Place> m(smem);
        Solver<Model, Place> s0(*m.a, tid, m.red, m.shs);
        s0.pack_blocks();
        pk0 = s0.pk[0];
        pk1 = s0.pk[1];
    }
    // Roles.  The LAST workgroup is the stepper: it owns no element, sweeps the score granules, forms the step and publishes
    // the next theta (with the error and convergence words) as granules of the same tag.  The others are workers: they
    // solve their elements, fetch the theta-free inputs of their first element of the next iteration (Prefetch: the loads
    // travel while the slowest worker -- the one with an element more -- is still solving, and through the step) and wait
    // for theta.  (With the step on every workgroup -- the first version -- the step's arrays needed the LDS that now
    // receives the prefetched normals.)
    // (round 5, LoopArgs::stepper_solves: with more elements than workers the stepper owns elements too -- 513 elements on 255
    //  workers are three rounds for three of them, on 256 solvers two rounds and the data element's shorter solve -- and sweeps
    //  once its own are done, the entries of a chunk requested together as on a node's board)
    const bool solving = ((kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr())[(sizeof(BatchArgs) + offsetof(LoopArgs, stepper_solves)) / 4] != 0;
    const int stepper_id = (int)gridDim.x - 1;
    const int nworkers = solving ? (int)gridDim.x : (int)gridDim.x - 1;   // the workgroups that own elements
    const bool stepper = (int)blockIdx.x == stepper_id;
    Prefetch<Place::EPT> pf;
    pf.p = -1;
    pf.have_n1 = pf.have_n2 = pf.have_x = pf.have_xg = pf.g_pending = false;
    typename Place::VZ zkeep;   // the MAP of the worker's last solve, carried in registers into the next iteration (see below)
    zkeep.clear();
    typedef __attribute__((address_space(1))) unsigned long long gu64;
#ifdef MUSE_STAMPS   // diagnostic build: the last iteration's times (100 MHz clock, comparable across the chip) of worker 0 (one
                     // of those with an element more), a worker in the middle and the stepper, behind the problems' rows
    auto loop_stamp = [&](int k) {
        LoopLds<Place> m(smem);
        const int b = (int)blockIdx.x, row = b == stepper_id ? 2 : (b == 0 ? 0 : (b == stepper_id / 2 ? 1 : -1));
        if (tid == 0 && m.a->stamps && row >= 0) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            m.a->stamps[(size_t)(m.a->nproblems + row) * 16 + k] = t;
        }
    };
#else
    auto loop_stamp = [](int) {};
#endif
    // (two loops, one per role -- round 5: in ONE loop with the roles as its branches, whatever a worker carries in registers from
    //  one iteration to the next -- the kept MAP -- was also live through the stepper's branch)
    // the per-iteration fields of the worker's LDS copy of the arguments, by ONE thread: iteration 1's before the loop, iteration
    // i + 1's while other lanes form the exponentials of its theta (one barrier and a serial stretch less between two iterations)
    auto iteration_setup = [&](int it) {
        LoopLds<Place> m(smem);
        BatchArgs& a = *m.a;   // (mutable here: theta and the per-iteration fields are re-written between iterations)
        const LoopArgs& L = *m.L;
        a.z0_mode = (it > 1 || L.z0_warm) ? Z0_WARM : Z0_ZERO;
        // every iteration re-draws the same streams at a new theta (src/muse.jl:134,169): the first one stores the
        // standard normals (or finds them: the host's mode), the later ones load them instead of running the generator
        if (it > 1 && a.ncache) a.ncache_mode = 2;
        a.scores = L.scores_out + (int64_t)(it - 1) * L.scores_stride;
        a.info = L.info_out + (int64_t)(it - 1) * L.info_stride;
        a.gran_tag = L.tag_base + (unsigned)it;
        if constexpr (Place::kXgLds) {  // the dummy slot and the pad element (N odd) hold 0 while problems run
            m.lds_x[a.ld] = 0.0;
            m.lds_g[a.ld] = 0.0;
            m.lds_x[a.ld + 1] = 0.0;
            m.lds_g[a.ld + 1] = 0.0;
            if (a.N < a.ld && pf.p < 0) {   // (a prefetched vector brings its pad element along; begin() masks it)
                m.lds_x[a.N] = 0.0;
                m.lds_g[a.N] = 0.0;
            }
        }
    };
    if (!stepper) {
    for (int iter = 1;; ++iter) {
        int err = STEP_OK, converged = 0;
        if (iter == 1 && tid == 0) iteration_setup(1);   // (the later iterations': beside the exponentials of the new theta, below)
        __syncthreads();
        {
            loop_stamp(0);
            {
                LoopLds<Place> m(smem);
                const BatchArgs& a = *m.a;
                // test hook (muse_debug_flags bit 4): every second worker leaves before its first solve -- a loop that dies with part of
                // the normals cache unwritten (tests/test_gpu_rows.py: the cache must not be taken for valid afterwards)
                if ((a.debug & 16) && ((int)blockIdx.x & 1)) return;
                double* wg_scratch = a.scratch + (int64_t)blockIdx.x * a.scratch_stride;
                // The worker's elements w, w + W, ... are visited in ALTERNATING order (round 5): upwards in the odd iterations,
                // downwards in the even ones, so that an iteration begins with the element the previous one ended with -- whose
                // MAP, the warm start it needs, is still in the z registers (Solver::run, keep_z: no load, no clear).  A
                // worker with ONE element (the per-GPU share of a sharded job) never loads a warm start at all.  The solves are
                // independent of one another, so the order changes no bit.  (MUSE_DEBUG bit 3: the old order, always reloading.)
                // (debug bit 6, a tuning aid: a stepper that solves takes the data element, problem 0, for itself and the simulations are
                //  dealt from problem 1 on -- see the stepper's loop)
                const int first = (int)blockIdx.x + ((solving && a.include_data && (a.debug & 64)) ? 1 : 0);
                const int cnt = m.L->deal_q + ((int)blockIdx.x < m.L->deal_r ? 1 : 0);   // >= 1: the host sizes the grid so
                const bool alternate = Solver<Model, Place>::kKeepZ && !(a.debug & 8);
                const bool up = !alternate || (iter & 1);
                const int step = up ? nworkers : -nworkers;
                int p = up ? first : first + (cnt - 1) * nworkers;
                // (ONE copy of the solve in the worker's loop -- round 5; the first problem used to be peeled off, a second copy of the
                //  whole solver in the kernel)
                for (int k = 0; k < cnt; ++k) {
                    Solver<Model, Place> sv(a, tid, m.red, m.shs);
                    sv.pk[0] = pk0;
                    sv.pk[1] = pk1;
                    const int nx = k + 1 < cnt ? p + step : -1;
                    bool kept = false;
                    if constexpr (Solver<Model, Place>::kKeepZ) {
                        if (__builtin_expect(k == 0 && alternate && iter > 1 && sv.can_keep(p), 1)) {
                            sv.z = zkeep;
                            sv.begin_kept(p, wg_scratch, m.lds_x, m.lds_g, pf);
                            kept = true;
                        }
                    }
                    if (!kept) sv.template begin<false, false>(p, wg_scratch, m.lds_x, m.lds_g, pf);
                    // Round 6: an iteration is bound by memory (four words per element and problem).  The LAST problem's
                    // MAP stays in registers for the next iteration's first solve, and nothing else reads its slot while the loop
                    // runs: it is not stored now (a word less for one problem in two -- for the only problem of a worker with one
                    // element: a rank's share, the reference's default 100 simulations) but when the loop ends, below.
                    if constexpr (Solver<Model, Place>::kKeepZ) {
                        if (k == cnt - 1 && alternate && iter < m.L->maxsteps && !(a.debug & 512) && sv.keeps_next(p)) sv.d.zslot = -1;
                    }
                    sv.after_begin(p, m.lds_x, m.lds_g, pf, nx);
                    if constexpr (Solver<Model, Place>::kKeepZ) zkeep = sv.z;   // (the last one's stays: the next iteration's first)
                    wg_barrier<!Model::kStencil>();   // (raw: the next problem's n1 is on its way into the g area)
                    if (k == 0) loop_stamp(1);
                    if (k + 1 < cnt) p += step;
                }
                loop_stamp(2);
                if constexpr (Place::kXgLds) {   // the next iteration's first problem: the one just solved (p), or the first again
                    if (iter < m.L->maxsteps && a.ncache_mode != 0 && !(a.debug & 4))
                        prefetch_issue<T>(a, tid, alternate ? p : first, m.lds_x, m.lds_g, pf, true);
                }
                loop_stamp(3);
            }
            // ---- wait for the stepper's granules: theta_next [nt], then {err, converged} as one double
            {
                LoopLds<Place> m(smem);
                BatchArgs& a = *m.a;
                const int nt = a.ntheta;
                const gu64* gran = (const gu64*)m.L->theta_gran;
                const unsigned tag = a.gran_tag;
                const int ngran = 2 * (nt + 1);
                unsigned* out = reinterpret_cast<unsigned*>(m.rec);   // (the record area is the stepper's; a worker parks theta here)
                if (tid < 64) {
                    int tl = tid;
                    asm volatile("" : "+v"(tl));
                    const bool live = tl < ngran;
                    unsigned long long gv = 0, t_wait0 = 0;
                    unsigned spins = 0;
                    for (;;) {
                        bool ok = true;
                        if (live) {
                            gv = __hip_atomic_load(gran + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = (unsigned)(gv >> 32) == tag;
                        }
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        __builtin_amdgcn_s_sleep(2);
                        if ((++spins & 0xffu) == 0) {   // bounded by TIME (4 s): workgroups that are not all resident must not hang the GPU
                            unsigned long long now;
                            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                            if (t_wait0 == 0) t_wait0 = now;
                            else if (now - t_wait0 > 400000000ull) {
                                __hip_atomic_store((gi32*)a.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                if (tl == 0) m.flags[2] = 1;
                                break;
                            }
                        }
                    }
                    if (live) out[tl] = (unsigned)(gv & 0xffffffffull);
                }
                __syncthreads();
                loop_stamp(4);
                const double* th = m.rec;
                const unsigned long long ec = (unsigned long long)__double_as_longlong(th[nt]);
                err = m.flags[2] ? (int)STEP_TIMEOUT : __builtin_amdgcn_readfirstlane((int)(ec & 0xffffffffull));
                converged = __builtin_amdgcn_readfirstlane((int)(ec >> 32));
                if (err != STEP_OK || converged || iter == m.L->maxsteps) {
                    if constexpr (Solver<Model, Place>::kKeepZ) {
                        // the loop ends before its last iteration: the MAP held back above (the same conditions, formed again rather
                        // than carried across the solves) goes to its slot
                        const bool alternate = !(a.debug & 8);
                        if (iter < m.L->maxsteps && alternate && !(a.debug & 512)) {
                            const int first = (int)blockIdx.x + ((solving && a.include_data && (a.debug & 64)) ? 1 : 0);
                            const int cnt = m.L->deal_q + ((int)blockIdx.x < m.L->deal_r ? 1 : 0);
                            const int last = (iter & 1) ? first + (cnt - 1) * nworkers : first;
                            Solver<Model, Place> sv(a, tid, m.red, m.shs);
                            if (sv.keeps_next(last)) {
                                sv.z = zkeep;
                                sv.store_kept(last);
                            }
                        }
                    }
                    break;
                }
                // a lane per exponential (step.hpp, make_map_theta_component's statements): exp(theta/2), exp(-theta) side by side, the
                // constant term and the next iteration's fields on lanes of their own
                int tl = tid;
                asm volatile("" : "+v"(tl));   // (else tid - kMaxTheta is formed at the kernel's entry and held -- spilled -- across the solves)
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
                if (tl <= kMaxTheta) {
                    pair_theta_update(tl, nt, a.bnd, th, a.cur);
                } else if (tl == 2 * kMaxTheta + 1) {
                    iteration_setup(iter + 1);
                }
#else
                if (tl < kMaxTheta) {
                    const bool live = tl < nt;
                    a.cur.t.theta[tl] = live ? th[tl] : 0.0;
                    a.cur.t.sd[tl] = live ? muse_exp(0.5 * th[tl]) : 0.0;
                } else if (tl < 2 * kMaxTheta) {
                    const int k = tl - kMaxTheta;
                    a.cur.t.iv[k] = k < nt ? muse_exp(-th[k]) : 0.0;
                } else if (tl == 2 * kMaxTheta) {
                    make_map_theta_const(nt, a.bnd, th, a.cur);
                } else if (tl == 2 * kMaxTheta + 1) {
                    iteration_setup(iter + 1);
                }
#endif
                // (the barrier at the top of the next iteration orders these writes before the first problem reads them)
            }
        }
    }
    } else {
    for (int iter = 1;; ++iter) {
        int err = STEP_OK, converged = 0;
        if (solving && iter == 1 && tid == 0) iteration_setup(1);
        __syncthreads();
        if (solving) {
            // the stepper's own elements, as a worker's -- without the kept MAP and the fetch across the step (this workgroup's LDS
            // is the step's between two iterations; it has the fewest elements of all and is not the one the others wait for)
            loop_stamp(0);
            LoopLds<Place> m(smem);
            const BatchArgs& a = *m.a;
            double* wg_scratch = a.scratch + (int64_t)blockIdx.x * a.scratch_stride;
            // (debug bit 6: the data element is the stepper's own and the workers hold simulations only -- at 512 of them on 256 compute
            //  units two each.  Measured at configs[1], tools/stamps_run.py: iterations whose solves are converged at the start 32.8 us
            //  against 39.2 with the data element dealt like the rest, but iterations whose solves take a line search -- every
            //  iteration of a run as short as muse()'s own -- 56 against 53: data + 2 simulations are 47 us on whichever workgroup,
            //  and a worker starts every second iteration with the data element's MAP still in its registers.  Not the default.)
            const bool own_data = a.include_data && (a.debug & 64);
            const int first = (int)blockIdx.x + (own_data ? 1 : 0);
            for (int p = own_data ? 0 : first; p < a.nproblems;) {
                const int nx = (own_data && p == 0) ? first : p + nworkers;
                Solver<Model, Place> sv(a, tid, m.red, m.shs);
                sv.pk[0] = pk0;
                sv.pk[1] = pk1;
                sv.template run<false>(p, wg_scratch, m.lds_x, m.lds_g, pf, nx < a.nproblems ? nx : -1);
                wg_barrier<!Model::kStencil>();
                p = nx;
            }
            loop_stamp(2);
            __syncthreads();   // (x and g are dead from here on: the step's arrays alias them)
        }
        {
            // ---- the stepper.  ONE wavefront does everything (the others wait at the barrier below): lane l takes the
            // simulations s = l, l + 64, ... -- it polls each score's two granules (one 16-byte load) until both carry this
            // iteration's tag, in the order in which its partial sum adds them, so that when the last element's score lands
            // one addition per lane, the tree and the few scalar operations of the step are all that is left to do.
            // stepbuf: gs [nprob][nt] (data element first), small[24] = {-, mean[8], var[8]}, StepWork
            LoopLds<Place> m(smem);
            BatchArgs& a = *m.a;
            const LoopArgs& L = *m.L;
            const int nt = a.ntheta, S = L.sp.nsims;
            const int64_t H = MUSE_RUN_HIST(nt);
            double* gs = m.stepbuf;
            double* small = gs + (int64_t)L.nprob_total * nt;   // (the whole job's elements: a sharded loop's rank solves a share of them)
            StepWork& w = *reinterpret_cast<StepWork*>(small + 24);
            if (tid == 0) {
                int zero = 0;
                asm volatile("" : "+v"(zero));   // (else a 64-bit zero is held across the stepper's own solves, in scratch)
                a.gran_tag = L.tag_base + (unsigned)iter;
                m.flags[0] = zero;
                m.flags[1] = zero;
                m.flags[2] = zero;
            }
            __syncthreads();
            if (tid < 64) {
                const unsigned tag = a.gran_tag;
                const rsrc_t grs = make_rsrc(L.score_gran, (int64_t)(2 * nt * L.nprob_total) * 8);
                bool expired = false;
                int lane = tid;
                asm volatile("" : "+v"(lane));
                // The node's board (muse_run_sharded's device loop): the scores of EVERY rank's elements, in pinned host memory.  A
                // poll is a PCIe round trip (~2 us), so the in-order poll below -- eight dependent polls per lane at 512 simulations
                // -- is replaced by batched sweeps: a lane's (up to) sixteen entries of a chunk of 1024 are requested together and
                // re-requested until all of them carry this iteration's tag; complete entries go to gs[] in LDS, from which the
                // sums below take them in the same order as ever.  After the last score has landed: one sweep.
                const bool board = __builtin_amdgcn_readfirstlane(L.board) != 0 || solving;   // (a stepper with elements of its own: no
                                                                                               //  poll was under way while it solved)
                if (board) {
                    const int nent = L.nprob_total * nt;
                    constexpr int kSweep = 16;   // entries a lane has in flight: 513 entries (512 simulations and the data, one component) are ONE chunk
                    for (int e0 = 0; e0 < nent; e0 += 64 * kSweep) {
                        unsigned pending = 0;
#pragma unroll
                        for (int j = 0; j < kSweep; ++j) pending |= (e0 + 64 * j + lane < nent) ? (1u << j) : 0u;
                        unsigned spins = 0;
                        unsigned long long t_wait0 = 0;
                        for (;;) {
                            double lo[kSweep], hi[kSweep];
#pragma unroll
                            for (int j = 0; j < kSweep; ++j) {
                                const int e = (pending >> j) & 1u ? e0 + 64 * j + lane : 0x08000000;   // (done or beyond the end: out of range, no access)
                                load_f64x2<kCoherent>(grs, 2 * e, lo[j], hi[j]);
                            }
#pragma unroll
                            for (int j = 0; j < kSweep; ++j) {
                                const unsigned long long glo = (unsigned long long)__double_as_longlong(lo[j]), ghi = (unsigned long long)__double_as_longlong(hi[j]);
                                if (((pending >> j) & 1u) && (unsigned)(glo >> 32) == tag && (unsigned)(ghi >> 32) == tag) {
                                    gs[e0 + 64 * j + lane] = __longlong_as_double((long long)((ghi << 32) | (glo & 0xffffffffull)));
                                    pending &= ~(1u << j);
                                }
                            }
                            if (__builtin_amdgcn_ballot_w64(pending != 0u) == 0ull || expired) break;
                            __builtin_amdgcn_s_sleep(1);
                            if ((++spins & 0x3fu) == 0) {   // bounded by TIME (4 s)
                                unsigned long long now;
                                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                                if (t_wait0 == 0) t_wait0 = now;
                                else if (now - t_wait0 > 400000000ull) {
                                    __hip_atomic_store((gi32*)a.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    expired = true;
                                }
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wavefront: its LDS writes are visible to its reads below)
                }
                auto score = [&](int row, int k) -> double {   // element `row`'s component k, once both of its granules are this iteration's
                    if (board) return gs[row * nt + k];
                    unsigned spins = 0;
                    unsigned long long t_wait0 = 0;
                    for (;;) {
                        double lo, hi;
                        load_f64x2<kCoherent>(grs, 2 * (row * nt + k), lo, hi);
                        const unsigned long long glo = (unsigned long long)__double_as_longlong(lo), ghi = (unsigned long long)__double_as_longlong(hi);
                        if (((unsigned)(glo >> 32) == tag && (unsigned)(ghi >> 32) == tag) || expired)
                            return __longlong_as_double((long long)((ghi << 32) | (glo & 0xffffffffull)));
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 0xffu) == 0) {   // bounded by TIME (4 s): workgroups that are not all resident must not hang the GPU
                            unsigned long long now;
                            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                            if (t_wait0 == 0) t_wait0 = now;
                            else if (now - t_wait0 > 400000000ull) {
                                __hip_atomic_store((gi32*)a.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                expired = true;
                            }
                        }
                    }
                };
                constexpr int MAXB = Model::MAXB;
                // step.hpp's step_moments_wave for every component at once, with the poll in its first pass: a lane asks for ALL
                // components of a simulation together (their granules are neighbours), so that after the last score has
                // landed nothing is left but the sums -- polled one component after the other, every component paid a
                // lane's eight dependent round trips again
                double mk[MAXB], vk[MAXB];
#pragma unroll
                for (int k = 0; k < MAXB; ++k) mk[k] = vk[k] = 0.0;
                for (int sidx = lane; sidx < S; sidx += 64) {
                    double v[MAXB];
                    if (board) {
#pragma unroll
                        for (int k = 0; k < MAXB; ++k) v[k] = k < nt ? gs[(int64_t)(1 + sidx) * nt + k] : 0.0;
                    } else {   // all components of the simulation: their loads in flight together, until every granule carries the tag
                        unsigned spins = 0;
                        unsigned long long t_wait0 = 0;
                        for (;;) {
                            bool all = true;
#pragma unroll
                            for (int k = 0; k < MAXB; ++k) {
                                double lo = 0.0, hi = 0.0;
                                if (k < nt) load_f64x2<kCoherent>(grs, 2 * ((1 + sidx) * nt + k), lo, hi);
                                const unsigned long long glo = (unsigned long long)__double_as_longlong(lo), ghi = (unsigned long long)__double_as_longlong(hi);
                                all = all && (k >= nt || ((unsigned)(glo >> 32) == tag && (unsigned)(ghi >> 32) == tag));
                                v[k] = __longlong_as_double((long long)((ghi << 32) | (glo & 0xffffffffull)));
                            }
                            if (all || expired) break;
                            __builtin_amdgcn_s_sleep(1);
                            if ((++spins & 0xffu) == 0) {   // bounded by TIME (4 s)
                                unsigned long long now;
                                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                                if (t_wait0 == 0) t_wait0 = now;
                                else if (now - t_wait0 > 400000000ull) {
                                    __hip_atomic_store((gi32*)a.error_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    expired = true;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < MAXB; ++k)
                        if (k < nt) {
                            gs[(int64_t)(1 + sidx) * nt + k] = v[k];
                            mk[k] += v[k];
                        }
                }
#pragma unroll
                for (int k = 0; k < MAXB; ++k)
                    if (k < nt) {
                        mk[k] = wave_total<false>(mk[k]);
                        mk[k] /= S;
                    }
                for (int sidx = lane; sidx < S; sidx += 64) {
#pragma unroll
                    for (int k = 0; k < MAXB; ++k)
                        if (k < nt) {
                            const double dlt = gs[(int64_t)(1 + sidx) * nt + k] - mk[k];
                            vk[k] += dlt * dlt;
                        }
                }
#pragma unroll
                for (int k = 0; k < MAXB; ++k)
                    if (k < nt) {
                        vk[k] = wave_total<false>(vk[k]);
                        vk[k] /= (S - 1);
                        if (lane == 0) {
                            small[8 + k] = mk[k];
                            small[16 + k] = vk[k];
                        }
                    }
                if (lane < nt) {
                    gs[lane] = score(0, lane);   // the data element's score
                    w.theta[lane] = a.cur.t.theta[lane];
                }
                if (__builtin_amdgcn_ballot_w64(expired) != 0ull && lane == 0) m.flags[2] = 1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wavefront: its LDS writes above are visible to its reads below)
                loop_stamp(4);
                // ---- the step, one lane per component (step.hpp: step_record's pieces; every lane the same statements the host
                //      loop runs for that component, the sums over components in component order)
                int e = m.flags[2] ? (int)STEP_TIMEOUT : (int)STEP_OK;
                if (e == STEP_OK) {
                    if (lane < nt) step_component(L.sp, lane, w.theta, gs, small + 8, small + 16, w.rec, w);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (__builtin_amdgcn_ballot_w64(lane < nt && !step_like_ok(nt, lane, w.rec)) != 0ull) e = STEP_SINGULAR_LIKE;
                }
                if (e == STEP_OK) {
                    bool ok = true;
                    if (lane < nt) ok = step_post_diag(nt, lane, w.rec, w);
                    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) e = STEP_SINGULAR_POST;
                }
                int cv = 0;
                if (e == STEP_OK) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane < nt) step_row(L.sp, lane, w.rec, w.theta_next, w);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // the test at the top of iteration iter + 1 > 2, on this record and the previous one (src/muse.jl:163-166)
                    if (iter >= 2 && iter < L.maxsteps) {
                        if (lane < nt) small[lane] = step_converged_term(nt, lane, w.rec, m.rec);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        double q = 0.0;
                        for (int k = 0; k < nt; ++k) q += small[k];
                        const int c = step_converged_from(q, L.sp.theta_rtol);
                        if (c < 0) e = STEP_DOMAIN;
                        cv = c > 0;
                    }
                }
                if (lane == 0) {
                    if (e == STEP_OK) {
                        unsigned long long now;
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                        w.rec[7 * nt + nt * nt] = (double)(now - *m.t_prev) * 1e-8;   // seconds of this iteration
                        *m.t_prev = now;
                    }
                    m.flags[0] = e;
                    m.flags[1] = cv;
                    // the words the workers wait for
                    w.theta_next[nt] = __longlong_as_double((long long)(((unsigned long long)(unsigned)cv << 32) | (unsigned)e));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < 2 * (nt + 1)) {   // theta_next [nt] and the {err, converged} word, two tagged granules each
                    gu64* gran = (gu64*)L.theta_gran;
                    const unsigned long long b = (unsigned long long)__double_as_longlong(w.theta_next[lane >> 1]);
                    const unsigned half = (lane & 1) ? (unsigned)(b >> 32) : (unsigned)(b & 0xffffffffull);
                    __hip_atomic_store(gran + lane, ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                loop_stamp(5);
                if (e == STEP_OK) {
                    for (int k = lane; k < (int)H; k += 64) {
                        const double v = w.rec[k];
                        L.hist_out[(int64_t)(iter - 1) * H + k] = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    for (int k = lane; k < (int)H; k += 64) m.rec[k] = w.rec[k];
                    if (lane < nt) L.theta_out[lane] = w.theta_next[lane];
                    if (L.scores_all_out) {   // the sharded loop: every rank's stepper holds every score -- the iteration's block, data element first
                        const int nent = L.nprob_total * nt;
                        for (int k = lane; k < nent; k += 64) L.scores_all_out[(int64_t)(iter - 1) * nent + k] = gs[k];
                    }
                }
                if (lane == 0) {
                    L.status[0] = e == STEP_OK ? iter : iter - 1;
                    L.status[1] = e;
                    L.status[2] = cv;
                }
                if (e == STEP_OK) {   // (the stepper's own copy of theta: the next record's -- and its next solves')
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
                    if (lane <= kMaxTheta) pair_theta_update(lane, nt, a.bnd, w.theta_next, a.cur);
#else
                    if (lane < kMaxTheta) make_map_theta_component(lane, nt, w.theta_next, a.cur);
                    if (lane == 0) make_map_theta_const(nt, a.bnd, w.theta_next, a.cur);
#endif
                    if (solving) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the pad slots of x and g may lie inside the step's arrays)
                        if (lane == 0) iteration_setup(iter + 1);
                    }
                }
            }
            __syncthreads();
            err = __builtin_amdgcn_readfirstlane(m.flags[0]);
            converged = __builtin_amdgcn_readfirstlane(m.flags[1]);
            if (err != STEP_OK || converged || iter == L.maxsteps) break;
        }
    }
    }
}


// ================================================================================================
// Launch shims.
// ================================================================================================
constexpr int kMaxDevices = 64;  // devices of one process (HIP device ordinals)
template <class Model, class Place, bool IMPLICIT = false>
static hipError_t launch_one(const LaunchShape& s, const BatchArgs& a, hipStream_t stream) {
    auto kern = map_score_kernel<Model, Place, IMPLICIT>;
    // per instantiation AND per device (the attribute belongs to the device's copy of the function): raised once, not per launch
    static size_t lds_allowed[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if (s.lds > (lds_allowed[dev] ? lds_allowed[dev] : (size_t)48 * 1024)) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds);
        if (e != hipSuccess) return e;
        lds_allowed[dev] = s.lds;
    }
    // The completion event of a result area rides on the dispatch itself (its completion signal) instead of following
    // it as a packet of its own, which the next launch would have to wait behind.
    hipExtLaunchKernelGGL(kern, dim3(s.grid), dim3(Place::T), s.lds, stream, nullptr, (hipEvent_t)s.done_event, 0, a);
    return hipGetLastError();
}
template <class Model>
hipError_t launch_place(const LaunchShape& s, const BatchArgs& a, hipStream_t st) {
    const int pl = s.place;
    if constexpr (Model::kStencil) {
        if (pl == P_C256 && s.lds_s) return launch_one<Model, PlaceStreaming<256, true, kStencilU, true, true>>(s, a, st);
        if (pl == P_C256) return launch_one<Model, PlaceStreaming<256, true, kStencilU, true>>(s, a, st);
        if (pl == P_S256) return launch_one<Model, PlaceStreaming<256, false, kStencilU>>(s, a, st);
        return launch_one<Model, PlaceStreaming<512, false, kStencilU>>(s, a, st);
    } else {
        switch (pl) {
            case P_R256x1: return launch_one<Model, PlaceResident<256, 1, false>>(s, a, st);
            case P_R512x4: return launch_one<Model, PlaceResident<512, 4, false>>(s, a, st);
            case P_R512x10: return launch_one<Model, PlaceResident<512, 10, true>>(s, a, st);
            case P_C256: return launch_one<Model, PlaceStreaming<256, true, kStreamU>>(s, a, st);
            case P_CR2: return launch_one<Model, PlaceResident<512, 5, false, true>>(s, a, st);
            case P_CR4: return launch_one<Model, PlaceResident<512, 3, false, true>>(s, a, st);
            case P_CR8: return launch_one<Model, PlaceResident<512, 2, false, true>>(s, a, st);
            case P_S256: return launch_one<Model, PlaceStreaming<256, false, kStreamU>>(s, a, st);
            default: return launch_one<Model, PlaceStreaming<512, false, kStreamU>>(s, a, st);
        }
    }
}
// The big tier (ntheta > kMaxTheta) runs in the streaming policy only.
template <class Model>
hipError_t launch_place_big(const LaunchShape& s, const BatchArgs& a, hipStream_t st) {
    if constexpr (Model::kStencil) {
        if (s.place == P_C256 && s.lds_s) return launch_one<Model, PlaceStreaming<256, true, kStencilU, true, true>>(s, a, st);
        if (s.place == P_C256) return launch_one<Model, PlaceStreaming<256, true, kStencilU, true>>(s, a, st);
        if (s.place == P_S512) return launch_one<Model, PlaceStreaming<512, false, kStencilU>>(s, a, st);
        if (s.place == P_S256) return launch_one<Model, PlaceStreaming<256, false, kStencilU>>(s, a, st);
    } else {
        if (s.place == P_C256) return launch_one<Model, PlaceStreaming<256, true, kStreamU>>(s, a, st);
        if (s.place == P_S512) return launch_one<Model, PlaceStreaming<512, false, kStreamU>>(s, a, st);
        if (s.place == P_S256) return launch_one<Model, PlaceStreaming<256, false, kStreamU>>(s, a, st);
    }
    return hipErrorInvalidValue;
}
// The implicit-differentiation H runs in the streaming policy only (single workgroup, or a cluster for large N).
template <class Model>
hipError_t launch_place_implicit(const LaunchShape& s, const BatchArgs& a, hipStream_t st) {
    constexpr int U = Model::kStencil ? kStencilU : kStreamU;
    if (s.place == P_C256) return launch_one<Model, PlaceStreaming<256, true, U, Model::kStencil>, true>(s, a, st);
    return launch_one<Model, PlaceStreaming<512, false, U>, true>(s, a, st);
}

// ---- the device-resident loop: one instantiation per (model, non-cluster placement) -------------------------------------
enum LoopOp { LOOP_QUERY, LOOP_GRID, LOOP_LAUNCH };
struct LoopCall {
    LoopOp op;
    const BatchArgs* a;
    const LoopArgs* l;
    hipStream_t st;
    int num_cus;
    int* max_grid;
    int* scratch_bytes;   // LOOP_GRID: the kernel's private segment per lane (what its spilled registers take)
};
template <class Model, class Place>
static hipError_t loop_one(const LaunchShape& s, const LoopCall& c) {
    auto kern = muse_loop_kernel<Model, Place>;
    if (c.op == LOOP_QUERY) return hipSuccess;
    static size_t lds_allowed[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if (s.lds > (lds_allowed[dev] ? lds_allowed[dev] : (size_t)48 * 1024)) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds);
        if (e != hipSuccess) return e;
        lds_allowed[dev] = s.lds;
    }
    if (c.op == LOOP_GRID) {
        int per_cu = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, Place::T, s.lds);
        if (e != hipSuccess) return e;
        *c.max_grid = per_cu * c.num_cus;
        if (c.scratch_bytes) {
            hipFuncAttributes at;
            const hipError_t e2 = hipFuncGetAttributes(&at, (const void*)kern);
            if (e2 != hipSuccess) return e2;
            *c.scratch_bytes = (int)at.localSizeBytes;
        }
        return hipSuccess;
    }
    hipLaunchKernelGGL(kern, dim3(s.grid), dim3(Place::T), s.lds, c.st, *c.a, *c.l);
    return hipGetLastError();
}
template <class Model>
hipError_t loop_place(const LaunchShape& s, const LoopCall& c) {
    // The resident placements only (round 5): muse_run_device routes every streaming placement and the stencil model to the host
    // loop (measured slower in loop form: N = 30 000 x 512 sims 377 against 347 us per iteration), so their loop kernels were
    // dead code that nothing launched and no test ran.
    if constexpr (!Model::kStencil) {
        switch (s.place) {
            case P_R256x1: return loop_one<Model, PlaceResident<256, 1, false>>(s, c);
            case P_R512x4: return loop_one<Model, PlaceResident<512, 4, false>>(s, c);
            case P_R512x10: return loop_one<Model, PlaceResident<512, 10, true>>(s, c);
            default: break;
        }
    }
    return hipErrorNotSupported;  // streaming and cluster placements: the host loop (muse_run) runs those
}

// Which models' kernels live in which translation unit (kernels_part.hip, -DMUSE_PART=n); muse_kernels.hip declares them `extern`.
#define MUSE_INSTANTIATE_ELEMENTWISE(X, M)                                                          \
    X template hipError_t launch_place<M>(const LaunchShape&, const BatchArgs&, hipStream_t);       \
    X template hipError_t launch_place_implicit<M>(const LaunchShape&, const BatchArgs&, hipStream_t); \
    X template hipError_t loop_place<M>(const LaunchShape&, const LoopCall&);
#define MUSE_INSTANTIATE_STENCIL(X, M)                                                              \
    X template hipError_t launch_place<M>(const LaunchShape&, const BatchArgs&, hipStream_t);       \
    X template hipError_t launch_place_implicit<M>(const LaunchShape&, const BatchArgs&, hipStream_t);
#define MUSE_INSTANTIATE_BIG(X, M)                                                                  \
    X template hipError_t launch_place_big<M>(const LaunchShape&, const BatchArgs&, hipStream_t);   \
    X template hipError_t launch_place_implicit<M>(const LaunchShape&, const BatchArgs&, hipStream_t);
#ifdef MUSE_USER_MODEL_HEADER
#ifdef MUSE_MODEL_SECOND
#define MUSE_INSTANTIATE_USER(X, M)                                                                 \
    X template hipError_t launch_place<M>(const LaunchShape&, const BatchArgs&, hipStream_t);       \
    X template hipError_t launch_place_implicit<M>(const LaunchShape&, const BatchArgs&, hipStream_t); \
    X template hipError_t loop_place<M>(const LaunchShape&, const LoopCall&);
#define MUSE_INSTANTIATE_USER_BIG(X, M) MUSE_INSTANTIATE_BIG(X, M)
#else
#define MUSE_INSTANTIATE_USER(X, M)                                                                 \
    X template hipError_t launch_place<M>(const LaunchShape&, const BatchArgs&, hipStream_t);       \
    X template hipError_t loop_place<M>(const LaunchShape&, const LoopCall&);
#define MUSE_INSTANTIATE_USER_BIG(X, M) X template hipError_t launch_place_big<M>(const LaunchShape&, const BatchArgs&, hipStream_t);
#endif
#ifdef MUSE_MODEL_PAIR   // two parameters per block: tiers of 2, 4 and 8 components (1, 2, up to 4 blocks); no implicit differentiation, no big tier
#define MUSE_INSTANTIATE_PAIR(X, M)                                                                 \
    X template hipError_t launch_place<M>(const LaunchShape&, const BatchArgs&, hipStream_t);       \
    X template hipError_t loop_place<M>(const LaunchShape&, const LoopCall&);
#define MUSE_PART_0(X) MUSE_INSTANTIATE_PAIR(X, UserModel<2>)
#define MUSE_PART_1(X) MUSE_INSTANTIATE_PAIR(X, UserModel<4>)
#define MUSE_PART_2(X) MUSE_INSTANTIATE_PAIR(X, UserModel<kMaxTheta>)
#else
#define MUSE_PART_0(X) MUSE_INSTANTIATE_USER(X, UserModel<1>)
#define MUSE_PART_1(X) MUSE_INSTANTIATE_USER(X, UserModel<kMaxTheta>)
#define MUSE_PART_2(X) MUSE_INSTANTIATE_USER_BIG(X, UserModel<kBigTheta>)
#endif
#define MUSE_PART_3(X)
#define MUSE_PART_4(X)
#define MUSE_PART_5(X)
#define MUSE_PART_6(X)
#define MUSE_PART_7(X)
#else
#define MUSE_PART_0(X) MUSE_INSTANTIATE_ELEMENTWISE(X, FunnelModel<1>)
#define MUSE_PART_1(X) MUSE_INSTANTIATE_ELEMENTWISE(X, FunnelModel<2>)
#define MUSE_PART_2(X) MUSE_INSTANTIATE_ELEMENTWISE(X, FunnelModel<4>)
#define MUSE_PART_3(X) MUSE_INSTANTIATE_ELEMENTWISE(X, FunnelModel<kMaxTheta>)
#define MUSE_PART_4(X) MUSE_INSTANTIATE_ELEMENTWISE(X, NoiseModel)
#define MUSE_PART_5(X) MUSE_INSTANTIATE_STENCIL(X, SmoothModel<2>) MUSE_INSTANTIATE_STENCIL(X, SmoothModel<4>)
#define MUSE_PART_6(X) MUSE_INSTANTIATE_STENCIL(X, SmoothModel<kMaxTheta>) MUSE_INSTANTIATE_BIG(X, SmoothModel<kBigTheta>)
#define MUSE_PART_7(X) MUSE_INSTANTIATE_BIG(X, FunnelModel<kBigTheta>)
#endif
constexpr int kKernelParts = 8;
}  // namespace muse
