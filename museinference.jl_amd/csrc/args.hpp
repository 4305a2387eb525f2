// args.hpp -- constants, the kernel argument block and the element -> problem map of the MUSE engine (see muse_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/muse_hip.h"

namespace muse {


constexpr int kM = 10;         // L-BFGS memory (Optim.LBFGS default m)
constexpr int kMaxIter = 1000;  // Optim.Options default iterations
constexpr int kMaxTheta = MUSE_MAX_THETA;
constexpr int kBigTheta = MUSE_MAX_THETA_EXT;  // ntheta in (kMaxTheta, kBigTheta]: the "big" tier (BigTheta below)
constexpr int kResultAreas = 4;
constexpr int kMaxCluster = 16;
constexpr int kClusterSlotDoubles = 2 * kMaxCluster * 8 * 2;  // two parities x members x 8 values x 2 granules
constexpr int64_t kClusterMinN = 65536;  // N >= this: several workgroups cooperate on one problem
constexpr int64_t kMaxResidentN = 10000;

// HagerZhang() defaults of LineSearches.jl
constexpr double kHzDelta = 0.1, kHzSigma = 0.9, kHzRho = 5.0, kHzEpsilon = 1e-6, kHzGamma = 0.66, kHzPsi3 = 0.1;
constexpr int kHzLinesearchMax = 50, kHzIterFiniteMax = 52;
constexpr double kEps = 2.220446049250313e-16;

struct ThetaSet {
    double theta[kMaxTheta];
    double sd[kMaxTheta];  // exp(theta/2)  (muse_exp: the fixed sequence of theta_math.hpp, the same bits on host and device)
    double iv[kMaxTheta];  // exp(-theta)
};
struct SampleSd {
    double sd[kMaxTheta];  // exp(theta/2) of a theta simulations are DRAWN at (it differs from the MAP's theta in get_H!'s maps)
};
// theta of a MAP problem and of its score, with the constant term of -2 logLike that goes with it
struct MapTheta {
    ThetaSet t;
    double f_const;  // sum_k N_k theta_k
    double pad_;
};
static_assert(sizeof(MapTheta) % 16 == 0, "copied in 4-byte words into a 16-byte aligned LDS block");
constexpr int kMaxMaps = 8;  // independent maps (each with a theta of its own) that ONE launch can carry
// ntheta > kMaxTheta: the per-block coefficients of the MAP's theta and the block boundaries, read by the kernel straight from
// the kernel-argument segment (they share the place of maps[]: such a launch carries one map); BatchArgs::cur then holds
// f_const only, and a finite-difference batch's sampling entries (tsample) are kBigTheta doubles each
struct BigTheta {
    double sd[kBigTheta], iv[kBigTheta];
};

enum { X_SAMPLE = 0, X_DATA = 1, X_GIVEN = 2 };
enum { Z0_ZERO = MUSE_Z0_ZERO, Z0_TRUE = MUSE_Z0_TRUE, Z0_WARM = MUSE_Z0_WARM, Z0_COPY = 3 };
enum { BATCH_STD = 0, BATCH_FD = 1, BATCH_SINGLE = 2, BATCH_IMPLICIT = 3 };

// Storage policies / launch shapes of the solver kernel (models.hpp has the types).  P_CRx: register-resident clusters of
// 2 / 4 / 8 workgroups of 512 threads with 5 / 3 / 2 pairs per thread (capacity 5120 / 6144 / 8192 pairs >=
// kMaxResidentN / 2), selected by muse_set_element_split.
enum PlaceId { P_S256 = 0, P_S512 = 1, P_R256x1 = 2, P_R512x4 = 3, P_R512x10 = 4, P_C256 = 5, P_CR2 = 6, P_CR4 = 7, P_CR8 = 8 };
#ifndef MUSE_STENCIL_U
#define MUSE_STENCIL_U 2
#endif
constexpr int kStencilU = MUSE_STENCIL_U;  // pairs per trip for the stencil model: 2 (no spills, 212 VGPRs; measured on smooth_1e5:
                                           // 4.65 ms against 4.76 at 3 and 4.80 at 4, which spills 25 registers)
#ifndef MUSE_STREAM_U
#define MUSE_STREAM_U 4
#endif
constexpr int kStreamU = MUSE_STREAM_U;    // pairs per trip of the elementwise models' streaming passes (a pure
                                           // performance knob: a thread visits its pairs in the same order for every U)

#ifndef MUSE_STREAM_GEN_U
#define MUSE_STREAM_GEN_U 1
#endif
constexpr int kStreamGenU = MUSE_STREAM_GEN_U;  // pairs per trip of the streaming placements' sampler pass (their normals
                                                // are drawn side by side: 2 * kStreamGenU generator chains per thread;
                                                // noise_1e6: 1.57 ms at 1, 1.71 ms at 2)
#ifndef MUSE_BG_U
#define MUSE_BG_U 2
#endif
#ifndef MUSE_BG_PAIRS
#define MUSE_BG_PAIRS 1
#endif
#ifndef MUSE_SAMPLER_PAIRS
#define MUSE_SAMPLER_PAIRS 1
#endif
constexpr int kSamplerPairs = MUSE_SAMPLER_PAIRS;  // pairs drawn per trip of the LDS-resident placement's sampler (1 or 2:
                                                   // two or four generator chains side by side per thread).  At two waves
                                                   // per SIMD the generator is bound by SIMD throughput, not by latency:
                                                   // measured 56.0 us (1) against 58.0 us (2) at configs[1]

struct BatchArgs {
    int64_t N, ld;
    int ntheta, kind;
    int64_t bnd[kMaxTheta + 1];  // block k = elements [bnd[k], bnd[k+1])
    int bnd32[kMaxTheta + 1];    // the same, 32-bit (N < 2^28), for the per-element block lookup
    int pad0_;
    uint64_t seed;
    double atol;
    int nproblems, include_data, z0_mode, store_zhat;
    int cg_maxiter;   // BATCH_IMPLICIT: IterativeSolvers.cg maxiter (reference default 100)
    int debug;        // profiling aids: bit0 skip the solve (sample + score only), bit1 take x from the data vector
    int64_t sim_begin, fid_slot, slot0;
    MapTheta cur;                  // theta of the MAP problem and of the score.  The LDS copy of this field is re-written at the
                                   // start of a problem when the launch carries several maps (maps[p / n_per_map]), and by the
                                   // step between two iterations of the device-resident muse! loop (muse_loop_kernel)
    int nmaps, n_per_map;          // BATCH_STD: problem p is element p % n_per_map of map p / n_per_map
    int64_t map_stride;            // score rows per map in the output block (>= n_per_map: a gathered block is padded)
    const SampleSd* tsample;       // FD: exp(theta/2) of the sampling thetas, [fd_grid * ntheta] (column j, grid point g at
                                   // j * fd_grid + g; shared by the simulations) or, fd_per_problem, one entry per problem
    int fd_grid, fd_per_problem;   // BATCH_FD: grid points per (simulation, column) unit (central_fdm(3,1): +step, -step = 2)
    const double* x_data;          // [ld]
    const double* x_given;         // BATCH_SINGLE: [ld]
    double* zhat;                  // [slots][ld]
    double* scores;                // [nproblems][ntheta]
    muse_info* info;               // [nproblems]
    double* scratch;               // per workgroup
    int64_t scratch_stride;        // doubles per workgroup
    int* work_counter;             // monotonically increasing ticket counter (never reset)
    int ticket_base;               // this launch's tickets are work_counter values base .. base+nproblems-1
    int p0;                        // BATCH_FD / BATCH_IMPLICIT: the batch starts at this element of sim_begin's own list
                                   // (a column range that begins inside a simulation's Jacobian)
    // cluster mode (several workgroups per problem): csize workgroups 0..csize-1 of cluster blockIdx/csize
    int csize, nclusters;
    double* cl_part;               // [nclusters][kClusterSlotDoubles]: partial sums / maxima, or their tagged granules
    unsigned int* cl_state;        // [nclusters] epoch reached by the cluster's granule exchange (persists across launches)
    int* error_flag;               // [0] set when a bounded cluster wait expires; [1] the largest epoch reported
    int xcd_local;                 // cluster mode: the members of a cluster are workgroups b = x (mod 8), which the dispatcher
                                   // places on ONE XCD (round-robin over the XCDs): their exchanges stay inside it
    unsigned long long* clock_out; // non-null (bench.py's roofline leg only): workgroup 0 stores {s_memtime, s_memrealtime} at its
                                   // entry ([0], [1]) and exit ([2], [3]): shader clock = d(memtime) / d(memrealtime) x 100 MHz
    unsigned long long* stamps;    // diagnostic build (-DMUSE_STAMPS) only: [nproblems][16] shader-clock stamps
    // Standard normals of simulation streams already drawn inside the SAME host call (muse_run's later
    // iterations re-draw every simulation at a new theta, the FD batch draws each simulation 2*ntheta times):
    // [slot][2][ld], slot = sim - ncache_sim0.  mode 1: generate and store; mode 2: load instead of generating.
    double* ncache;
    int64_t ncache_sim0;
    int ncache_count, ncache_mode;
    int nstd;  // BATCH_STD: elements >= nstd only draw (and store) the normals of sim norm_sim0 + (p - nstd)
    int imp_split;  // BATCH_IMPLICIT: elements per simulation (1: all H columns in one element; ntheta: one each)
    int64_t norm_sim0;
    // device-resident muse! loop (muse_loop_kernel): every element also publishes its score components as tagged 8-byte
    // granules {32-bit half, 32-bit tag} -- [row * ntheta + k][2], one write-through store each -- which every workgroup
    // sweeps at the end of the iteration (no fence, no barrier: solver.hpp, cluster_exchange, has the argument)
    unsigned long long* gran;      // (the sharded loop: this rank's rows of the NODE's score board -- pinned host memory that every
                                   //  rank's GPU maps -- and gran_sys = 1: the stores are system-scope)
    unsigned int gran_tag;         // this iteration's tag
    int gran_sys;
    // BATCH_FD with the fiducial MAP folded into the launch (round 6; muse_engine.cpp, fd_values_impl): problem 0 is the ONE fiducial
    // MAP of src/muse.jl:417-423 (simulation fid_sim at theta0 from zero(z), stored to slot fid_slot), problems 1.. are the
    // perturbed ones; each of those draws its x -- which does not need the fiducial -- and then waits for fid_flag to carry
    // fid_tag before it loads its warm start (the fiducial's workgroup releases its stores and sets the flag: solver.hpp, run)
    int fd_fold;
    unsigned int fid_tag;
    int64_t fid_sim;
    unsigned int* fid_flag;
    int64_t pad3_;
    union {  // read from the kernarg segment only (never copied to LDS)
        alignas(16) MapTheta maps[kMaxMaps];  // theta of every map, nmaps > 1
        BigTheta big;                         // ntheta > kMaxTheta
    };
    // LAST, kernarg segment only: the run-time constants of a user-supplied model (include/muse_model.h, muse_const) -- the
    // launching context's device vectors and their lengths.  Per LAUNCH (round 5; a process-wide __device__ symbol before:
    // a launch of another context of the same library still in flight would have read the new owner's pointers).
    const double* consts[4];
    long const_len[4];
    // kernarg segment only: gran_sys == 2 -- the sharded loop with a board per GPU in DEVICE memory (muse_comm.cpp: every rank's board
    // is mapped into every rank by hipIpc): an element's score granules are stored into EVERY rank's board (posted writes over xGMI),
    // each at this rank's row offset; gran itself is not used then
    unsigned long long* gran_peers[8];
    int ngran_peers, pad2_;
};
constexpr size_t kArgsConstsOffset = offsetof(BatchArgs, consts);
static_assert(offsetof(BatchArgs, const_len) == kArgsConstsOffset + 4 * sizeof(const double*), "muse_const reads {pointers[4], lengths[4]}");
static_assert(sizeof(BigTheta) <= sizeof(MapTheta) * kMaxMaps, "the big tier's tables take the place of maps[]");
constexpr size_t kArgsHeadBytes = offsetof(BatchArgs, maps);  // what the kernel keeps in LDS
static_assert(kArgsHeadBytes % 16 == 0 && offsetof(BatchArgs, cur) % 8 == 0, "LDS copy of the argument block");
static_assert(sizeof(BatchArgs) <= 4096, "kernarg segment");

struct ProblemDesc {
    int64_t sim;
    int64_t row;       // row of the element's score in the output block
    int nslot;         // slot of the simulation's normals in the cache, -1: none
    bool normals_only;
    int x_mode, z0_mode, tsample;  // tsample < 0: sample at tmap
    int64_t zslot, z0slot;         // zslot < 0: zhat not stored
};

#ifdef __HIPCC__
__device__ __forceinline__ ProblemDesc describe(const BatchArgs& a, int p) {
    ProblemDesc d;
    d.normals_only = false;
    d.row = p;
    if (a.kind == BATCH_STD && a.nmaps > 1) {  // several independent maps in one launch: slots and infos by p, scores by (map, element)
        const int m = p / a.n_per_map, e = p - m * a.n_per_map;
        const bool data = a.include_data && e == 0;
        d.sim = data ? -1 : a.sim_begin + e - (a.include_data ? 1 : 0);
        d.x_mode = (data || (a.debug & 2)) ? X_DATA : X_SAMPLE;
        d.z0_mode = (data && a.z0_mode == Z0_TRUE) ? Z0_ZERO : a.z0_mode;
        d.tsample = -1;
        d.zslot = a.store_zhat ? a.slot0 + p : -1;
        d.z0slot = a.slot0 + p;
        d.row = (int64_t)m * a.map_stride + e;
    } else if (a.kind == BATCH_STD && p >= a.nstd) {
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
    } else if (a.kind == BATCH_FD && a.fd_fold && p == 0) {   // the launch's own fiducial MAP: results behind the perturbed problems'
        d.sim = a.fid_sim;
        d.x_mode = X_SAMPLE;
        d.z0_mode = Z0_ZERO;
        d.tsample = -1;
        d.zslot = a.fid_slot;
        d.z0slot = a.fid_slot;
        d.row = a.nproblems - 1;   // (solver.hpp, info_row: the info likewise)
    } else if (a.kind == BATCH_FD) {
        if (a.fd_fold) {
            p -= 1;
            d.row = p;
        }
        const int per = a.fd_grid * a.ntheta, pp = p + a.p0;
        d.sim = a.sim_begin + pp / per;
        d.x_mode = X_SAMPLE;
        d.z0_mode = Z0_COPY;
        d.tsample = a.fd_per_problem ? p : pp % per;
        d.zslot = -1;
        d.z0slot = a.fid_slot >= 0 ? a.fid_slot : a.slot0 + pp / per;
    } else if (a.kind == BATCH_IMPLICIT) {
        d.sim = a.sim_begin + (p + a.p0) / (a.imp_split > 1 ? a.imp_split : 1);
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
#endif  // __HIPCC__

// what the step needs to know beside the scores (the plain option set of muse_run, include/muse_hip.h)
struct StepParams {
    int ntheta, nsims;
    int prior_kind;  // 0 flat, 1 independent Gaussian
    int pad_;
    double alpha, theta_rtol;
    double prior_mean[kMaxTheta], prior_sigma[kMaxTheta];
};
// Arguments of the device-resident muse! loop beside the map's own (muse_kernels.hip: muse_loop_kernel): ONE launch runs
// every outer iteration -- map, exchange of the scores between the workgroups, step (step.hpp), next map.
struct LoopArgs {
    StepParams sp;
    int maxsteps, z0_warm;
    unsigned int tag_base;           // the granule tag of iteration i is tag_base + i (grows from run to run: nothing is reset)
    int board;                       // 1: score_gran is the node's score board in pinned host memory (muse_run_sharded's device loop: the
                                     // ranks' workers write their scores there, every rank's stepper polls ALL of them -- batched
                                     // sweeps, a PCIe round trip each -- and takes the same step from the same bits)
    const unsigned long long* score_gran;  // what the stepper polls: [nprob_total * ntheta][2] tagged granules, data element first
    unsigned long long* theta_gran;  // this GPU's own: the stepper's theta_next [ntheta] and {err, converged}, two granules each
    int nprob_total;                 // elements of the WHOLE job (nsims + 1); BatchArgs::nproblems is this rank's share
    int stepper_solves;              // 1: the last workgroup owns elements like the others and steps when its own are solved (more
                                     // elements than workgroups: a workgroup that only steps would cost the others a round)
    int deal_q, deal_r;              // the deal of the elements (the host's division): workgroup w owns deal_q + (w < deal_r) of them
    int64_t scores_stride;           // doubles between two iterations' score blocks at scores_out (0: one block, overwritten)
    double* scores_all_out;          // board mode: pinned [maxsteps][nprob_total][ntheta], written by the stepper
    double* hist_out;                // pinned [maxsteps][MUSE_RUN_HIST]
    double* scores_out;              // pinned [maxsteps][nsims + 1][ntheta]: every iteration's scores, data element first
    muse_info* info_out;             // pinned [maxsteps][nsims + 1], or a device dummy of one iteration (info_stride 0)
    int64_t info_stride;             // nsims + 1, or 0
    double* theta_out;               // pinned [ntheta]: the iterate after the last executed step
    int* status;                     // pinned: [0] iterations executed, [1] STEP_* error, [2] converged
};

// launch shims of muse_kernels.hip (the only translation unit that holds device code)
struct LaunchShape {
    int model, ntheta, place, grid;
    bool implicit;
    bool big;    // the big tier's instantiation (BigTheta): ntheta > kMaxTheta, or 2..kMaxTheta components of an elementwise model in a
                 // streaming placement (muse_engine.cpp, tier_big: the same bits, 2.4x faster than the small tiers' streaming passes)
    bool lds_s;  // stencil model in a cluster: the search direction in LDS (vec.hpp, LdsMirror)
    size_t lds;
    void* done_event;  // hipEvent_t (or null) that the launch itself signals on completion: no separate event packet
};
hipError_t launch_solver(const LaunchShape& s, const BatchArgs& a, hipStream_t stream);
hipError_t launch_sample(int model, const BatchArgs& a, uint64_t sim, double* x, double* z, double* noise, hipStream_t stream);
// the standard normals of ONE stream into a slot of the normals cache's layout ([2][ld]: n1, n2), drawn by the whole GPU instead of
// by the one workgroup that solves the stream's problem (muse_engine.cpp, fd_values_impl: the fiducial MAP of get_H!)
hipError_t launch_normals(uint64_t seed, uint64_t sim, int64_t ld, double* slot, hipStream_t stream);
hipError_t launch_loglike(int model, const BatchArgs& a, const double* x, const double* z, double* g, double* out, hipStream_t stream);
// the device-resident loop: false where the placement has no loop kernel (cluster placements); max_grid = the number of
// workgroups that are certainly resident at once (they meet at the end of every iteration)
bool loop_supported(const LaunchShape& s);
hipError_t loop_max_grid(const LaunchShape& s, int num_cus, int* max_grid, int* scratch_bytes);
hipError_t launch_loop(const LaunchShape& s, const BatchArgs& a, const LoopArgs& l, hipStream_t stream);
size_t loop_extra_lds(bool xg_lds, int64_t nprob, int ntheta);  // LDS of the loop kernel beside the map kernel's
size_t loop_step_bytes(int64_t nprob, int ntheta);              // the step's own arrays (they alias x and g in the LDS-resident layout)
constexpr int kArgsDoubles = (int)((kArgsHeadBytes + 15) / 16 * 2);  // LDS copy of the kernel arguments (without the trailing maps[])


}  // namespace muse
