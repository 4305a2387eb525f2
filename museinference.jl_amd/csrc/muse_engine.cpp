// muse_engine.cpp -- host side of the MI355X engine for the MUSE inner loop: context, workspace, launch geometry, the
// C ABI of include/muse_hip.h and the native muse! outer loop.  All device code is in muse_kernels.hip; this file is
// compiled as plain C++ (a change here rebuilds in seconds, not in the minutes the kernel instantiations take).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/muse_hip.h"
#include "args.hpp"
#include "step.hpp"
#include "switches.hpp"
#include "user_model.hpp"

// ================================================================================================
// Host side: context, workspace, launches, C ABI.
// ================================================================================================
using namespace muse;

#if defined(__x86_64__) || defined(__i386__)
#define MUSE_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define MUSE_CPU_RELAX() asm volatile("yield" ::: "memory")
#else
#define MUSE_CPU_RELAX() std::this_thread::yield()
#endif
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

// What a solver launch owns exclusively while it runs.  A context has one set of these per LANE; the lane a call uses is
// swapped into the context's own fields of the same names (use_lane), so that the launch code below is written once.
// One lane (the default): launches of a context run one after the other on its stream.  muse_set_concurrency(ctx, n):
// result area r runs on lane r mod n -- a stream, a workgroup scratch, a ticket counter and a cluster state of its own --
// so that a launch can start on the compute units the previous one has already left, instead of behind its last workgroup
// and a launch gap.
constexpr int kMaxLanes = 4;
struct LaneState {
    hipStream_t stream = nullptr;
    double* scratch = nullptr;
    size_t scratch_doubles = 0;
    int* counter = nullptr;
    unsigned int ticket_base = 0;
    double* cl_part = nullptr;
    unsigned int* cl_state = nullptr;
    int cl_cap = 0;
    int* error_flag = nullptr;
    double* zhat = nullptr;        // the lane's resident MAP slots: a streaming solve works IN its slot, so maps in flight at once
    int64_t zhat_slots = 0;        // cannot share them
    bool ready = false;
};

struct muse_ctx {
    LaneState lanes[kMaxLanes];   // lanes[cur_lane] is stale while that lane is swapped in
    int nlanes = 1, cur_lane = 0;
    int model = 0, ntheta = 1, device = 0, placement = -1, num_cus = 0;
    int64_t N = 0, ld = 0;
    int64_t bnd[kBigTheta + 1] = {0};
    hipStream_t stream = nullptr, own_stream = nullptr;
    double* x_data = nullptr;
    bool has_data = false;
    double* zhat = nullptr;
    int64_t zhat_slots = 0;
    double* scratch = nullptr;
    size_t scratch_doubles = 0;
    int* counter = nullptr;
    double* tmp = nullptr;  // 3 vectors for the per-sim operator entry points
    SampleSd* tsample_dev = nullptr;   // sampling thetas of a finite-difference map (grown on demand)
    SampleSd* tsample_pin = nullptr;
    size_t tsample_cap = 0;
    // result areas: device + pinned host, each [cap] scores and infos
    double* scores_dev[kResultAreas] = {nullptr};
    muse_info* info_dev[kResultAreas] = {nullptr};
    double* scores_pin[kResultAreas] = {nullptr};
    muse_info* info_pin[kResultAreas] = {nullptr};
    int64_t res_cap[kResultAreas] = {0};
    int64_t res_n[kResultAreas] = {0};     // infos (elements) of the area's last launch
    int64_t res_rows[kResultAreas] = {0};  // score rows of it (>= res_n: a gathered multi-map block is padded per map)
    bool area_inflight[kResultAreas] = {false};  // launched, not yet waited for
    bool area_failed[kResultAreas] = {false};    // was in flight when a cluster wait expired (check_error_flag)
    double* small_dev = nullptr;  // 16 doubles
    hipEvent_t ev0 = nullptr, ev1 = nullptr, last0 = nullptr, last1 = nullptr;
    bool ev_valid = false;
    hipEvent_t area_done[kResultAreas] = {nullptr};  // recorded after an area's launch: its results are complete (kernel-end release)
    // live kernel timing: a ring of event pairs, one per solver launch (muse_profile_*)
    std::vector<hipEvent_t> prof_ev;
    int prof_count = 0;
    bool prof_on = false;
    double* ncache = nullptr;            // normals cache [ncache_slots][2][ld] (muse_run, FD batches, repeated maps)
    int64_t ncache_slots = 0;
    // what the cache holds: the standard normals of simulations [nc_sim0, nc_sim0 + nc_count) of master seed nc_seed (none:
    // nc_count == 0), and the simulation range of the last plain map (a range is cached when it is asked for AGAIN)
    uint64_t nc_seed = 0, nc_seen_seed = 0;
    int64_t nc_sim0 = 0, nc_count = 0, nc_seen_sim0 = 0, nc_seen_count = 0;
    // run-time constants of a user model (include/muse_model.h: muse_const): this context's device vectors and host copies
    double* consts_dev[4] = {nullptr, nullptr, nullptr, nullptr};
    double* consts_host[4] = {nullptr, nullptr, nullptr, nullptr};
    long consts_len[4] = {0, 0, 0, 0};
    bool has_consts = false;
    bool nc_auto = true;                 // muse_set_normals_cache: plain maps may store / load the normals of repeated simulations
    double* cl_part = nullptr;           // [cl_cap][kClusterSlotDoubles]
    unsigned int* cl_state = nullptr;    // [cl_cap] granule-exchange epochs
    int cl_cap = 0;
    int* error_flag = nullptr;           // pinned, device-mapped
    double* fid_norm = nullptr;          // [2][ld]: the standard normals of get_H!'s fiducial stream, drawn by a kernel of its own (fd_values_impl)
    unsigned int* fid_flag = nullptr;    // device word: the tag of the last fiducial MAP published inside a finite-difference launch
    unsigned int fid_tag = 0;            // (fd_values_impl; grows with every such launch)
    int debug = 0;                       // muse_debug_flags
    Switches sw;                         // the environment switches as muse_ctx_create found them (switches.hpp)
    int split = 0;                 // muse_set_element_split: 0 = by N alone, >= 2 = workgroups per element
    unsigned int ticket_base = 0;  // value of the device ticket counter when the next launch starts
    hipEvent_t launch_done = nullptr;  // set around launch_batch: the event this launch signals when it completes
    bool launch_done_used = false;
    bool timing = false;           // record an event pair around every solver launch (muse_set_timing; costs ~12 us per launch)
    unsigned long long* stamps = nullptr;
    int64_t stamps_cap = 0;
    unsigned long long* clock_pin = nullptr;  // pinned [4]: {s_memtime, s_memrealtime} at entry and exit of workgroup 0 (profiling launches)
    struct RunBuffers* run = nullptr;  // buffers of the device-resident muse! loop (muse_run_device)
    bool loop_unfit = false;   // a loop kernel of this context could not keep its workgroups resident (a shared GPU): the later
                               // calls of muse_run_device go straight to muse_run instead of spinning into the bounded waits again
    int comm_reserve_cus = 0;  // compute units left to a device-side collective that runs beside the solver (muse_comm.cpp)
    void* comm = nullptr;  // ncclComm_t (muse_comm.cpp)
    double* comm_buf = nullptr;
    size_t comm_buf_doubles = 0;
};

#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
// A library of the two-parameter family (include/muse_model.h, MUSE_MODEL_PAIR): K = ntheta / 2 blocks; block k's parameters are
// theta[k] and theta[K + k], its coefficients and constant come from the HEADER (muse_model_coefs, evaluated here on the host) and
// sit side by side in the tables: ThetaSet::sd and ::iv read as one [block][4] array, a sampling entry as [block][2] (csrc/models.hpp).
constexpr bool kPairModel = true;
static void pair_map_theta(int nt, const int64_t* bnd, const double* theta, MapTheta& m) {
    const int K = nt / 2;
    memset(&m, 0, sizeof m);
    double cst = 0.0;
    for (int k = 0; k < K; ++k) {
        double cf[4] = {0.0, 0.0, 0.0, 0.0};
        const double C = muse_model_coefs(theta[k], theta[K + k], cf);
        m.t.theta[k] = theta[k];
        m.t.theta[K + k] = theta[K + k];
        double* rec = &m.t.sd[0] + 4 * k;   // (csrc/models.hpp, pair_table: ThetaSet::sd and ::iv as ONE [block][4] table)
        for (int j = 0; j < 4; ++j) rec[j] = cf[j];
        cst += (double)(bnd[k + 1] - bnd[k]) * C;
    }
    m.f_const = cst;
}
#else
constexpr bool kPairModel = false;
#endif
static int nblocks_of(int ntheta) { return kPairModel ? ntheta / 2 : ntheta; }
// exp(theta/2), exp(-theta) and the constant term: step.hpp's fixed sequences (the device-resident loop forms the same bits)
static void make_thetaset(const muse_ctx* c, const double* theta, ThetaSet& t) {
    MapTheta m;
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
    pair_map_theta(c->ntheta, c->bnd, theta, m);
#else
    make_map_theta(c->ntheta, c->bnd, theta, m);
#endif
    t = m.t;
}
static double theta_const(const muse_ctx* c, const double* theta) {
    MapTheta m;
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
    pair_map_theta(c->ntheta, c->bnd, theta, m);
#else
    make_map_theta(c->ntheta, c->bnd, theta, m);
#endif
    return m.f_const;
}

static bool place_is_cluster(int pl) { return pl >= P_C256; }
// test hook (MUSE_DEBUG_LOOP_OVERSUBSCRIBE / muse_debug_flags bit 18): a loop launch with more workgroups than can be resident at once
static bool oversubscribe(const muse_ctx* c) { return c->sw.loop_oversubscribe || (c->debug & kDebugLoopOversubscribe) != 0; }

// Cluster size: a function of N alone (results must not depend on how many problems share a launch), unless the
// caller asked for a split (muse_set_element_split: results then depend on (N, split), still not on the launch).
// Stencil model in a cluster of `csize` workgroups: LDS bytes of the workgroup's own elements of the search direction
// (vec.hpp, LdsMirror: pairs per thread rounded up to whole trips); it is kept there when two workgroups per CU fit.
static size_t stencil_lds_s_bytes(const muse_ctx* c, int csize) {
    const int64_t pairs = (c->ld / 2 + (int64_t)csize * 256 - 1) / ((int64_t)csize * 256);
    const int64_t cap = (pairs + kStencilU - 1) / kStencilU * kStencilU;
    return (size_t)cap * 256 * 16;
}
static bool stencil_lds_s(const muse_ctx* c, int csize) {
    return !c->sw.no_lds_s && c->model == MUSE_MODEL_SMOOTH && csize >= 2 && stencil_lds_s_bytes(c, csize) <= 72 * 1024;
}
static int cluster_size(const muse_ctx* c) {
    if (c->split >= 2) return c->split;
    if (c->sw.cluster_size > 0) return c->sw.cluster_size;  // tuning aid
    // stencil model: clusters of 16 where that lets the search direction live in LDS (N <= ~147 000)
    if (c->model == MUSE_MODEL_SMOOTH && c->N >= kClusterMinN && c->N < 4194304 && stencil_lds_s(c, 16)) return 16;
    return c->N >= 4194304 ? 16 : (c->N >= kClusterMinN ? 8 : 1);  // 8: smooth_1e5 2.80 ms (4: 3.22), noise_1e6 1.56 (4: 1.61)
}
static bool use_cluster(const muse_ctx* c) { return c->split >= 2 || c->N >= kClusterMinN; }

// Workgroup size is a function of N alone (256 threads for N <= 512, else 512), so that the
// streaming and the resident policy reduce in the same order and give bitwise equal results.
static int choose_place(const muse_ctx* c) {
    const bool small = c->N <= 512;
    // cluster mode: for the stencil model the neighbours owned by other workgroups become visible through
    // the agent-scope release/acquire of the cluster reduction that ends every pass (pass_barrier where a
    // pass has no reduction)
    const bool big = c->ntheta > kMaxTheta;  // the big tier (args.hpp, BigTheta) runs in the streaming placements
    if (c->split >= 2 && c->split <= 8 && c->model != MUSE_MODEL_SMOOTH && c->placement != 0 && c->N <= kMaxResidentN && !big)
        return c->split == 2 ? P_CR2 : (c->split == 4 ? P_CR4 : P_CR8);
    if (use_cluster(c)) return P_C256;
    if (c->model == MUSE_MODEL_SMOOTH || c->placement == 0 || c->N > kMaxResidentN || big) return small ? P_S256 : P_S512;
    if (small) return P_R256x1;
    if (c->N <= 4096) return P_R512x4;
    return P_R512x10;
}
// Which instantiation a launch of placement `pl` gets: the big tier's (args.hpp, BigTheta: tables from the kernarg segment, blocks
// by arithmetic, block sums in chunks) for more than kMaxTheta components -- and for 2..kMaxTheta components of an elementwise
// model whenever the placement streams and the launch carries one map: the results are the small tier's bit for bit (the same
// per-thread order, the same trees), the streaming passes much cheaper (funnel, N = 10^5, 8 components, 128 sims: 0.68 -> 0.28 ms;
// the small tier's compare chain and selected accumulations in every trip).  The stencil model gains nothing (2.32 / 2.38 ms).
static bool tier_big(const muse_ctx* c, int pl, int nmaps) {
    if (c->ntheta > kMaxTheta) return true;
    if (kPairModel) return false;   // (one tier: models.hpp, UserModel of the two-parameter family)
    return !c->sw.no_big_tier && c->ntheta > 1 && c->model != MUSE_MODEL_SMOOTH && nmaps <= 1 && (pl == P_S256 || pl == P_S512 || pl == P_C256);
}
static bool ncache_applies(const muse_ctx* c) {
    return !c->sw.no_ncache && choose_place(c) == P_R512x10;
}
static int place_threads(int pl) { return (pl == P_S256 || pl == P_R256x1 || pl == P_C256) ? 256 : 512; }
// workgroups per CU the grid is sized from (cluster placements: every member must be resident at once, and the
// register budget of the 512-thread resident kernels admits one workgroup per CU for certain, two only sometimes)
static int place_wgs_per_cu(int pl) {
    return (pl == P_S256 || pl == P_R256x1) ? 4 : ((pl == P_R512x10 || pl >= P_CR2) ? 1 : 2);
}
static size_t place_lds(const muse_ctx* c, int pl) {
    size_t fixed = (size_t)(2 * (place_threads(pl) / 64) * 8 + 42 + kArgsDoubles) * sizeof(double);
    if (place_is_cluster(pl)) fixed += (size_t)kMaxCluster * 8 * sizeof(double);
    if (pl == P_C256 && stencil_lds_s(c, cluster_size(c))) fixed += stencil_lds_s_bytes(c, cluster_size(c));
    if (pl == P_R512x10) fixed += (size_t)2 * (c->ld + 2) * sizeof(double);
    return fixed;
}
// streaming: x, g, s, z, the history, one extra vector; clusters: a second (x, s) pair for the background generator
static int64_t place_scratch_vectors(int pl) {
    return (pl == P_S256 || pl == P_S512) ? 4 + 2 * kM + 1 : pl == P_C256 ? 4 + 2 * kM + 1 + 2 : 2 * kM;
}

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
    c->nc_count = 0;
    if (hipMalloc(&c->ncache, (size_t)slots * 2 * c->ld * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        static bool warned = false;
        if (!warned) {  // the results are the same bits either way; only the generator runs again
            fprintf(stderr, "[libmuse_hip] normals cache of %lld slots not allocated: simulations are re-drawn by the generator\n",
                    (long long)slots);
            warned = true;
        }
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
static bool ncache_holds(const muse_ctx* c, uint64_t seed, int64_t sim0, int64_t count) {
    return c->ncache && c->nc_count > 0 && c->nc_seed == seed && sim0 >= c->nc_sim0 && sim0 + count <= c->nc_sim0 + c->nc_count;
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

// Fill the common fields and launch the solver for `a.nproblems` elements.
static int launch_batch(muse_ctx* c, BatchArgs& a) {
    a.N = c->N;
    a.ld = c->ld;
    a.ntheta = c->ntheta;
    for (int k = 0; k <= kMaxTheta; ++k) {
        a.bnd[k] = c->bnd[k];
        a.bnd32[k] = k < nblocks_of(c->ntheta) ? (int)c->bnd[k] : 0x7fffffff;
    }
    a.x_data = c->x_data;
    if (!a.zhat) a.zhat = c->zhat;
    a.work_counter = c->counter;
    if (a.nmaps <= 1) {
        a.nmaps = 1;
        a.n_per_map = a.nproblems;
        a.map_stride = a.nproblems;
    }
    a.debug = c->debug & 0xffff;   // (bits 16-21 are host-side: switches.hpp)
    a.stamps = (c->stamps && a.nproblems <= c->stamps_cap) ? c->stamps : nullptr;
    a.clock_out = c->prof_on ? c->clock_pin : nullptr;  // roofline leg only
    const bool implicit = a.kind == BATCH_IMPLICIT;
    const int pl = implicit ? (use_cluster(c) ? P_C256 : P_S512) : choose_place(c);
    int grid = c->num_cus * place_wgs_per_cu(pl);
    a.csize = 1;
    // cluster placements need every member resident at once: a device-side collective of the previous step (RCCL transport:
    // its kernel holds compute units on a high-priority stream until the slowest peer arrives) keeps a share of the GPU
    if (place_is_cluster(pl) && c->comm_reserve_cus > 0 && c->num_cus > 2 * c->comm_reserve_cus)
        grid = (c->num_cus - c->comm_reserve_cus) * place_wgs_per_cu(pl);
    if (place_is_cluster(pl) && c->nlanes > 1) grid /= c->nlanes;   // ... and so do the launches of the other lanes
    if (place_is_cluster(pl)) {
        // ... and so do the launches of OTHER PROCESSES on this GPU, which the engine cannot see: MUSE_SHARED_GPU_RANKS=n says that n
        // processes share the device (the development set-up of bench.py's gloo mode and of the multi-process tests: eight ranks
        // on one GPU -- without it their cluster launches starve each other of compute units until the bounded waits expire)
        const int sharers = c->sw.shared_gpu_ranks;
        if (sharers > 1) grid = grid / sharers > 0 ? grid / sharers : 1;
    }
    a.nclusters = 0;
    if (place_is_cluster(pl)) {
        // every workgroup of a cluster must be resident at once (they wait for each other): the grid is sized from
        // a workgroup count per CU that the kernel's register budget always admits
        a.csize = cluster_size(c);
        int ncl = grid / a.csize;
        if (ncl > a.nproblems) ncl = a.nproblems;
        if (ncl < 1) ncl = 1;
        a.nclusters = ncl;
        grid = ncl * a.csize;
        const bool wrap = c->error_flag[1] > 0x30000000;  // granule tags (epoch numbers) are 32-bit: start over in time
        if (ncl > c->cl_cap || wrap) {
            HIPCHK(hipStreamSynchronize(c->stream));
            const int cap = ncl > c->cl_cap ? ncl : c->cl_cap;
            if (c->cl_part) HIPCHK(hipFree(c->cl_part));
            if (c->cl_state) HIPCHK(hipFree(c->cl_state));
            c->cl_part = nullptr; c->cl_state = nullptr; c->cl_cap = 0;
            HIPCHK(hipMalloc(&c->cl_part, (size_t)cap * kClusterSlotDoubles * sizeof(double)));
            HIPCHK(hipMalloc(&c->cl_state, (size_t)cap * sizeof(unsigned int)));
            HIPCHK(hipMemsetAsync(c->cl_part, 0, (size_t)cap * kClusterSlotDoubles * sizeof(double), c->stream));
            HIPCHK(hipMemsetAsync(c->cl_state, 0, (size_t)cap * sizeof(unsigned int), c->stream));
            c->error_flag[1] = 0;
            c->cl_cap = cap;
        }
        a.cl_part = c->cl_part;
        a.cl_state = c->cl_state;
        // XCD-local clusters (muse_kernels.hip) for the elementwise models, whose members meet in scalar exchanges only:
        // 22.2 -> 21.3 us at 64 sims split 4, noise_1e6 1.555 -> 1.52 ms.  The stencil model, whose members also stream
        // each other's boundary elements, measured slower with all of a cluster's traffic in one XCD (2.35 -> 2.47 ms).
        a.xcd_local = (!c->sw.no_xcd_local && ncl % 8 == 0 && c->model != MUSE_MODEL_SMOOTH) ? 1 : 0;
    } else {
        if (grid > a.nproblems) grid = a.nproblems;
        if (grid < 1) grid = 1;
    }
    a.error_flag = c->error_flag;
    a.scratch_stride = place_scratch_vectors(pl) * c->ld;
    int rc = ensure_scratch(c, (size_t)(place_is_cluster(pl) ? a.nclusters : grid) * a.scratch_stride);
    if (rc) return rc;
    a.scratch = c->scratch;
    const size_t lds = place_lds(c, pl);
    // Tickets: a workgroup's first problem is its own index, every further one a ticket (problem grid + ticket), and
    // every workgroup draws exactly one ticket past the batch, so a launch advances the counter by exactly
    // (nproblems - grid) + grid = nproblems (grid <= nproblems) -- no per-launch memset.  Wrap-around: reset explicitly.
    if (c->ticket_base > 0x70000000u) {
        HIPCHK(hipMemsetAsync(c->counter, 0, 16, c->stream));
        c->ticket_base = 0;
    }
    a.ticket_base = (int)c->ticket_base;
    if (!place_is_cluster(pl)) c->ticket_base += (unsigned)a.nproblems;  // clusters draw no tickets
    hipEvent_t e0 = c->ev0, e1 = c->ev1;
    if (c->prof_on && (size_t)(2 * c->prof_count + 1) < c->prof_ev.size()) {
        e0 = c->prof_ev[2 * c->prof_count];
        e1 = c->prof_ev[2 * c->prof_count + 1];
        c->prof_count += 1;
    }
    const bool timed = c->timing || c->prof_on;
    if (timed) HIPCHK(hipEventRecord(e0, c->stream));
    {
        LaunchShape shape;
        shape.model = c->model; shape.ntheta = c->ntheta; shape.place = pl; shape.grid = grid; shape.implicit = implicit; shape.lds = lds;
        shape.big = tier_big(c, pl, a.nmaps);
        shape.lds_s = !implicit && pl == P_C256 && stencil_lds_s(c, a.csize);
        shape.done_event = c->launch_done;
        c->launch_done_used = c->launch_done != nullptr;
        const hipError_t e = launch_solver(shape, a, c->stream);
        if (e != hipSuccess) rc = fail(MUSE_ERR_HIP, std::string("solver launch: ") + hipGetErrorString(e));
    }
    if (rc) return rc;
    if (timed) {
        HIPCHK(hipEventRecord(e1, c->stream));
        c->last0 = e0;
        c->last1 = e1;
        c->ev_valid = true;
    }
    return MUSE_OK;
}

// Swap lane `l` into the context's launch-state fields (and the current one back into its slot); a lane's stream, ticket
// counter and error word are created on first use.
static int use_lane(muse_ctx* c, int l) {
    if (l == c->cur_lane) return MUSE_OK;
    LaneState& out = c->lanes[c->cur_lane];
    out.stream = c->stream; out.scratch = c->scratch; out.scratch_doubles = c->scratch_doubles; out.counter = c->counter;
    out.ticket_base = c->ticket_base; out.cl_part = c->cl_part; out.cl_state = c->cl_state; out.cl_cap = c->cl_cap;
    out.error_flag = c->error_flag; out.zhat = c->zhat; out.zhat_slots = c->zhat_slots; out.ready = true;
    LaneState& in = c->lanes[l];
    if (!in.ready) {
        HIPCHK(hipStreamCreateWithFlags(&in.stream, hipStreamNonBlocking));
        HIPCHK(hipMalloc(&in.counter, 16));
        HIPCHK(hipMemset(in.counter, 0, 16));
        HIPCHK(hipHostMalloc(&in.error_flag, 64, hipHostMallocDefault));
        in.error_flag[0] = in.error_flag[1] = 0;
        in.ready = true;
    }
    c->stream = in.stream; c->scratch = in.scratch; c->scratch_doubles = in.scratch_doubles; c->counter = in.counter;
    c->ticket_base = in.ticket_base; c->cl_part = in.cl_part; c->cl_state = in.cl_state; c->cl_cap = in.cl_cap;
    c->error_flag = in.error_flag;
    c->zhat = in.zhat;
    c->zhat_slots = in.zhat_slots;
    c->cur_lane = l;
    return MUSE_OK;
}
template <class F>
static int for_each_lane(muse_ctx* c, F&& f) {   // f() with every lane that exists swapped in; lane 0 afterwards
    int rc = MUSE_OK;
    for (int l = 0; l < kMaxLanes && rc == MUSE_OK; ++l) {
        if (l != c->cur_lane && !c->lanes[l].ready) continue;
        rc = use_lane(c, l);
        if (rc == MUSE_OK) rc = f();
    }
    const int rc0 = use_lane(c, 0);
    return rc ? rc : rc0;
}

// A result area that is launched again (or whose pinned block an FD / implicit-diff map is about to reuse: those are
// hard-wired to area 1 on lane 0) while its previous launch is still in flight on ANOTHER lane: with one stream, stream order
// serialised the two; with lanes nothing does, and ensure_results / ensure_tsample / ensure_ncache only drain the current
// lane.  Drain every lane first (never taken by a caller that waits for an area before launching on it again).
static int settle_area(muse_ctx* c, int area) {
    if (!c->area_inflight[area] || c->nlanes <= 1) return MUSE_OK;
    return for_each_lane(c, [&]() -> int {
        HIPCHK(hipStreamSynchronize(c->stream));
        return MUSE_OK;
    });
}

#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_NCONST)
// The model's run-time constants as its functions see them (include/muse_model.h).  On the DEVICE every launch carries the
// launching context's pointers in its own argument block (BatchArgs::consts: set_launch_constants below) -- no state is shared
// between launches, so contexts of one library with different constants may have maps in flight at the same time.  On the HOST
// (muse_model_eval, the contract check of muse_ctx_create) the functions read these: ONE set per process and library, the set
// of the context that entered the library last (check_ctx); host evaluation is synchronous, nothing is in flight across it.
extern "C" {
const double* muse_host_consts[MUSE_MODEL_MAX_CONST] = {nullptr, nullptr, nullptr, nullptr};
long muse_host_const_len[MUSE_MODEL_MAX_CONST] = {0, 0, 0, 0};
}
static muse_ctx* g_consts_owner = nullptr;
static void own_host_constants(muse_ctx* c) {
    if (!c->has_consts || g_consts_owner == c) return;
    for (int k = 0; k < MUSE_MODEL_MAX_CONST; ++k) {
        muse_host_consts[k] = c->consts_host[k];
        muse_host_const_len[k] = c->consts_len[k];
    }
    g_consts_owner = c;
}
#define MUSE_OWN_CONSTANTS(c) own_host_constants(c)
#else
#define MUSE_OWN_CONSTANTS(c) do { } while (0)
#endif
// the launching context's constants into a launch's argument block (all zero for a model without any: muse_const is never called)
static void set_launch_constants(const muse_ctx* c, BatchArgs& a) {
    for (int k = 0; k < 4; ++k) {
        a.consts[k] = c->consts_dev[k];
        a.const_len[k] = c->consts_len[k];
    }
}

extern "C" {

const char* muse_last_error(void) { return g_err.c_str(); }
int64_t muse_max_resident_n(void) { return kMaxResidentN; }

const char* muse_model_name(int model) {
#ifdef MUSE_USER_MODEL_HEADER
    return model == MUSE_MODEL_USER ? MUSE_MODEL_NAME : nullptr;
#else
    static const char* const names[3] = {"funnel", "noise", "smooth"};
    return model >= 0 && model <= 2 ? names[model] : nullptr;
#endif
}

int muse_ctx_create(int model, int64_t N, int ntheta, int device, muse_ctx** out) {
    if (!out) return fail(MUSE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (model < 0 || model > MUSE_MODEL_USER) return fail(MUSE_ERR_INVALID, "unknown model");
    if (!muse_model_name(model))
        return fail(MUSE_ERR_INVALID, model == MUSE_MODEL_USER
                                          ? "MUSE_MODEL_USER: this library holds the built-in models only (build one from a model "
                                            "header, include/muse_model.h)"
                                          : "this library was built from a user's model header and holds MUSE_MODEL_USER only");
    if (N < 1) return fail(MUSE_ERR_INVALID, "N must be >= 1");
    if (ntheta < 1 || ntheta > kBigTheta) return fail(MUSE_ERR_INVALID, "ntheta must be in [1, MUSE_MAX_THETA_EXT]");
    if (model == MUSE_MODEL_NOISE && ntheta != 1) return fail(MUSE_ERR_INVALID, "MUSE_MODEL_NOISE has ntheta = 1");
    if (ntheta > N) return fail(MUSE_ERR_INVALID, "ntheta must be <= N");
    if (model == MUSE_MODEL_SMOOTH && N < 5) return fail(MUSE_ERR_INVALID, "MUSE_MODEL_SMOOTH needs N >= 5");
#ifdef MUSE_USER_MODEL_HEADER
#ifdef MUSE_MODEL_N  // a model with per-element tables is built for one N (include/muse_model.h)
    if (N != (int64_t)(MUSE_MODEL_N)) return fail(MUSE_ERR_INVALID, "model " MUSE_MODEL_NAME " was built for another N (MUSE_MODEL_N)");
#endif
#ifdef MUSE_MODEL_PAIR
    if (ntheta % 2 != 0 || ntheta > kMaxTheta)
        return fail(MUSE_ERR_INVALID, "model " MUSE_MODEL_NAME " has two parameters per block (MUSE_MODEL_PAIR): ntheta must be even and <= MUSE_MAX_THETA");
    if (ntheta / 2 > N) return fail(MUSE_ERR_INVALID, "more blocks than elements");
    {   // the pad element's contract of the two-parameter family: zero coefficients, x = z = 0 -> no contribution anywhere
        const double zero4[4] = {0.0, 0.0, 0.0, 0.0};
        double acc = 1.25, t0 = 1.0, t1 = 1.0;
        const double g0 = muse_model_grad(zero4, 0.0, 0.0, &acc, (long)N);
        muse_model_score_terms(zero4, 0.0, 0.0, &t0, &t1, (long)N);
        if (!(g0 == 0.0 && acc == 1.25 && t0 == 0.0 && t1 == 0.0))
            return fail(MUSE_ERR_INVALID, "model " MUSE_MODEL_NAME ": with all four coefficients 0 and x = z = 0, muse_model_grad must return 0 and "
                                          "leave acc unchanged and muse_model_score_terms must give 0, 0 (include/muse_model.h)");
    }
#else
    {   // the zero-element requirements of include/muse_model.h (the pad element of an odd-length vector must not contribute)
        double acc = 1.25;
        const double g0 = muse_model_grad(0.7, 0.0, 0.0, &acc, (long)N);
        if (!(g0 == 0.0 && acc == 1.25 && muse_model_score_term(0.0, 0.0, (long)N) == 0.0))
            return fail(MUSE_ERR_INVALID, "model " MUSE_MODEL_NAME ": muse_model_grad(iv, 0, 0, &acc, N) must return 0 and leave acc "
                                          "unchanged, muse_model_score_term(0, 0, N) must be 0 (include/muse_model.h)");
#ifdef MUSE_MODEL_SECOND
        double s4[4];
        muse_model_second(0.7, 0.0, 0.0, &s4[0], &s4[1], &s4[2], &s4[3], (long)N);
        if (!(isfinite(s4[0]) && isfinite(s4[1]) && isfinite(s4[2]) && isfinite(s4[3])))
            return fail(MUSE_ERR_INVALID, "model " MUSE_MODEL_NAME ": muse_model_second(iv, 0, 0, ..., N) must be finite (include/muse_model.h)");
#endif
    }
#endif
#endif
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MUSE_ERR_HIP, "no HIP device available (libmuse_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(MUSE_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    muse_ctx* c = new muse_ctx();
    c->sw = Switches::from_environment();   // the ONE place the environment is read (switches.hpp)
    c->model = model;
    c->N = N;
    c->ld = (N + 1) & ~(int64_t)1;
    c->ntheta = ntheta;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    {
        const int nb = nblocks_of(ntheta);   // (two parameters per block: ntheta / 2 blocks)
        for (int k = 0; k <= kBigTheta; ++k) {
            const int kk = k < nb ? k : nb;
            c->bnd[k] = ((int64_t)kk * N + nb - 1) / nb;
        }
    }
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(hipMalloc(&c->x_data, (size_t)c->ld * sizeof(double)));
    HIPCHK(hipMalloc(&c->counter, 16));
    HIPCHK(hipMemset(c->counter, 0, 16));
    HIPCHK(hipMalloc(&c->fid_flag, 64));
    HIPCHK(hipMemset(c->fid_flag, 0, 64));
    HIPCHK(hipHostMalloc(&c->error_flag, 64, hipHostMallocDefault));
    c->error_flag[0] = c->error_flag[1] = 0;
    HIPCHK(hipHostMalloc(&c->clock_pin, 64, hipHostMallocDefault));
    memset(c->clock_pin, 0, 64);
    HIPCHK(hipMalloc(&c->tmp, (size_t)3 * c->ld * sizeof(double)));
    HIPCHK(hipMalloc(&c->small_dev, (size_t)(1 + kBigTheta) * sizeof(double)));
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
static void free_run_buffers(muse_ctx* c);

// accessors for muse_comm.cpp (the context layout is private to this file)
int muse_set_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }
int muse_ctx_comm_slot(muse_ctx* c, void*** comm, int* device, void** stream) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    // the communicator's stream order (RCCL transport: collective stream <- event on the solver's stream) is tied to lane 0:
    // hand out lane 0's stream whatever lane the last map used
    const int rc = use_lane(c, 0);
    if (rc) return rc;
    *comm = &c->comm;
    *device = c->device;
    *stream = (void*)c->stream;
    return MUSE_OK;
}
int muse_ctx_switches(muse_ctx* c, const Switches** sw, int* debug) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    if (sw) *sw = &c->sw;
    if (debug) *debug = c->debug;
    return MUSE_OK;
}
int muse_ctx_set_comm_reserve(muse_ctx* c, int cus) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    c->comm_reserve_cus = cus > 0 ? cus : 0;
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
    (void)for_each_lane(c, [&]() { hipStreamSynchronize(c->stream); return MUSE_OK; });
    for (int l = 1; l < kMaxLanes; ++l) {   // lane 0's resources are the context's own fields now (freed below)
        LaneState& ln = c->lanes[l];
        if (!ln.ready) continue;
        hipFree(ln.scratch); hipFree(ln.counter); hipFree(ln.cl_part); hipFree(ln.cl_state); hipHostFree(ln.error_flag);
        hipFree(ln.zhat);
        hipStreamDestroy(ln.stream);
    }
    muse_comm_destroy(c);
    free_run_buffers(c);
    hipFree(c->cl_part); hipFree(c->cl_state); hipHostFree(c->error_flag); hipHostFree(c->clock_pin);
    hipFree(c->ncache);
    for (int k = 0; k < 4; ++k) {
        if (c->consts_dev[k]) hipFree(c->consts_dev[k]);
        free(c->consts_host[k]);
    }
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_NCONST)
    if (g_consts_owner == c) {
        g_consts_owner = nullptr;
        for (int k = 0; k < MUSE_MODEL_MAX_CONST; ++k) { muse_host_consts[k] = nullptr; muse_host_const_len[k] = 0; }
    }
#endif
    hipFree(c->x_data); hipFree(c->zhat); hipFree(c->scratch); hipFree(c->counter); hipFree(c->fid_flag); hipFree(c->fid_norm); hipFree(c->tmp);
    hipFree(c->small_dev); if (c->tsample_dev) hipFree(c->tsample_dev); if (c->tsample_pin) hipHostFree(c->tsample_pin);
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
    const int rc = use_lane(c, 0);   // every entry point works on lane 0 unless it says otherwise (map_async_impl)
    if (rc) return rc;
    MUSE_OWN_CONSTANTS(c);           // (a user model's run-time constants: this context's, if another one installed its own since)
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
int muse_set_element_split(muse_ctx* c, int split) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    if (split != 0 && split != 1 && split != 2 && split != 4 && split != 8 && split != 16)
        return fail(MUSE_ERR_INVALID, "split must be 0 (by N alone), 1, 2, 4, 8 or 16");
    if (split > kMaxCluster) return fail(MUSE_ERR_INVALID, "split exceeds the largest cluster");
    c->split = split == 1 ? 0 : split;
    return MUSE_OK;
}
// A bounded cluster wait that expired inside a solver kernel leaves the pinned error word set; every entry point that has
// synchronised with a launch reports it instead of returning the launch's garbage.  Recovery: the members that gave up went
// on with garbage reductions, so their control flow -- and their epoch counters -- diverged from their cluster's; only rank
// 0's epoch is saved, and granules tagged ahead of it are still lying in cl_part, where a later launch could take one for
// the current epoch's value.  So: drain the stream, start the cluster state over (tags and epochs from zero, as after the
// 32-bit wrap), and mark every result area whose launch was in flight as failed -- each of their waits reports the error
// once -- before the flag is cleared.
static int check_error_flag(muse_ctx* c) {
    bool any = *c->error_flag != 0;
    for (int l = 0; l < kMaxLanes; ++l)
        if (l != c->cur_lane && c->lanes[l].ready && c->lanes[l].error_flag[0]) any = true;
    if (!any) return MUSE_OK;
    (void)for_each_lane(c, [&]() -> int {
        (void)hipStreamSynchronize(c->stream);
        if (c->cl_part && c->cl_cap > 0) {
            (void)hipMemsetAsync(c->cl_part, 0, (size_t)c->cl_cap * kClusterSlotDoubles * sizeof(double), c->stream);
            (void)hipMemsetAsync(c->cl_state, 0, (size_t)c->cl_cap * sizeof(unsigned int), c->stream);
            (void)hipStreamSynchronize(c->stream);
        }
        c->error_flag[1] = 0;
        c->error_flag[0] = 0;
        return MUSE_OK;
    });
    for (int r = 0; r < kResultAreas; ++r) {
        if (c->area_inflight[r]) c->area_failed[r] = true;
        c->area_inflight[r] = false;
    }
    // a launch that raised the error word may have left its share of the normals cache unwritten: nothing is held any more
    c->nc_count = 0;
    c->nc_seen_count = 0;
    return fail(MUSE_ERR_HIP, "a cluster wait expired inside the solver kernel (workgroups of a cluster were not co-resident); "
                              "the cluster state was reset, result areas in flight are marked failed");
}
int muse_placement_info(muse_ctx* c, int* threads, int* workgroups_per_element, int* resident, int* direction_in_lds) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    const int pl = choose_place(c);
    const int cs = place_is_cluster(pl) ? cluster_size(c) : 1;
    if (threads) *threads = place_threads(pl);
    if (workgroups_per_element) *workgroups_per_element = cs;
    if (resident) *resident = (pl == P_R256x1 || pl == P_R512x4 || pl == P_R512x10 || pl >= P_CR2) ? 1 : 0;
    if (direction_in_lds) *direction_in_lds = (pl == P_C256 && stencil_lds_s(c, cs)) ? 1 : 0;
    return MUSE_OK;
}
int muse_synchronize(muse_ctx* c) {
    int rc = check_ctx(c);
    if (rc) return rc;
    return for_each_lane(c, [&]() -> int {
        HIPCHK(hipStreamSynchronize(c->stream));
        return MUSE_OK;
    });
}
int muse_set_constants(muse_ctx* c, int k, const double* values, int64_t count, int mem) {
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_NCONST)
    int rc = check_ctx(c);
    if (rc) return rc;
    if (k < 0 || k >= MUSE_MODEL_NCONST) return fail(MUSE_ERR_INVALID, "constant index out of range (MUSE_MODEL_NCONST of the model's header)");
    if (!values || count != c->N) return fail(MUSE_ERR_INVALID, "a constant vector has one entry per element (count == N)");
    rc = muse_synchronize(c);   // nothing in flight reads the vector that is about to be replaced
    if (rc) return rc;
    if (!c->consts_dev[k]) {
        if (hipMalloc(&c->consts_dev[k], (size_t)count * sizeof(double)) != hipSuccess) return fail(MUSE_ERR_ALLOC, "hipMalloc(constants) failed");
        c->consts_host[k] = (double*)malloc((size_t)count * sizeof(double));
        if (!c->consts_host[k]) return fail(MUSE_ERR_ALLOC, "malloc(constants) failed");
    }
    HIPCHK(hipMemcpyAsync(c->consts_dev[k], values, (size_t)count * sizeof(double), in_kind(mem), c->stream));
    HIPCHK(hipMemcpyAsync(c->consts_host[k], values, (size_t)count * sizeof(double),
                          mem == MUSE_MEM_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < count; ++i)
        if (!isfinite(c->consts_host[k][i])) {
            c->consts_len[k] = 0;
            return fail(MUSE_ERR_INVALID, "constants must be finite");
        }
    c->consts_len[k] = (long)count;
    c->has_consts = true;
    g_consts_owner = nullptr;   // (the host copies are re-read below)
    own_host_constants(c);
    return MUSE_OK;
#else
    (void)c; (void)k; (void)values; (void)count; (void)mem;
    return fail(MUSE_ERR_INVALID, "this library's model declares no run-time constants (MUSE_MODEL_NCONST, include/muse_model.h)");
#endif
}
int muse_model_has_second(void) {
#if defined(MUSE_USER_MODEL_HEADER) && (!defined(MUSE_MODEL_SECOND) || defined(MUSE_MODEL_PAIR))
    return 0;
#else
    return 1;
#endif
}
int muse_model_eval(muse_ctx* c, double iv, double sd, double x, double z, double n1, double n2, int64_t i, double* out) {
#ifdef MUSE_USER_MODEL_HEADER
#ifdef MUSE_MODEL_NCONST
    int rc = check_ctx(c);   // (the model's run-time constants become this context's)
    if (rc) return rc;
#else
    (void)c;                 // a model without run-time constants needs no context (and no GPU) for this
#endif
    if (!out || i < 0) return fail(MUSE_ERR_INVALID, "bad argument");
#ifdef MUSE_MODEL_PAIR
    // the two-parameter family: iv and sd are the block's two PARAMETERS here (a, b); out = {grad, objective term, t0, c0..c3 at
    // [3..6], z, x, the block's constant C, t1, 0}
    {
        double cf[4] = {0.0, 0.0, 0.0, 0.0}, acc = 0.0;
        out[9] = muse_model_coefs(iv, sd, cf);
        out[0] = muse_model_grad(cf, x, z, &acc, (long)i);
        out[1] = acc;
        muse_model_score_terms(cf, x, z, &out[2], &out[10], (long)i);
        for (int k = 0; k < 4; ++k) out[3 + k] = cf[k];
        muse_model_sample(cf, n1, n2, &out[7], &out[8], (long)i);
        out[11] = 0.0;
        return MUSE_OK;
    }
#else
    double acc = 0.0;
    out[0] = muse_model_grad(iv, x, z, &acc, (long)i);
    out[1] = acc;
    out[2] = muse_model_score_term(x, z, (long)i);
    muse_model_sample(sd, n1, n2, &out[7], &out[8], (long)i);
#ifdef MUSE_MODEL_SECOND
    muse_model_second(iv, x, z, &out[3], &out[4], &out[5], &out[6], (long)i);
    out[9] = muse_model_dx_dsd(sd, n1, n2, (long)i);
#else
    out[3] = out[4] = out[5] = out[6] = out[9] = NAN;
#endif
    return MUSE_OK;   // (out[0..9] only: callers of the one-parameter family pass ten doubles, as before round 6)
#endif
#else
    (void)c; (void)iv; (void)sd; (void)x; (void)z; (void)n1; (void)n2; (void)i; (void)out;
    return fail(MUSE_ERR_INVALID, "muse_model_eval evaluates a user-supplied model's header; this library holds the built-in models");
#endif
}
int muse_set_normals_cache(muse_ctx* c, int enabled) {
    if (!c) return fail(MUSE_ERR_INVALID, "ctx is NULL");
    c->nc_auto = enabled != 0;
    if (!c->nc_auto) c->nc_seen_count = 0;
    return MUSE_OK;
}
int muse_set_concurrency(muse_ctx* c, int nlanes) {
    int rc = muse_synchronize(c);
    if (rc) return rc;
    if (nlanes < 1 || nlanes > kMaxLanes) return fail(MUSE_ERR_INVALID, "concurrency must be in [1, 4]");
    c->nlanes = nlanes;
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
// Diagnostics (include/muse_hip.h).  Bits 0-8 travel to the kernels: 0 skip the solve, 1 x from the data vector, 2 the loop kernel
// does not prefetch, 3 its old element order, 4 test hook (odd workers leave), 5 no speculating trials, 6 a solving stepper takes the
// data element for itself, 7 the stepper never solves, 8 the data vector does not travel through the g area; bits 16-19 are
// host-side (switches.hpp): 16 host board, 17 host-driven sharded loop, 18 oversubscribed loop launch (test hook), 19 run timing,
// 20 the finite-difference map carries its fiducial MAP, 21 get_H!'s fiducial MAP draws its own normals.
int muse_debug_flags(muse_ctx* c, int flags) {
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
    rc = for_each_lane(c, [&]() -> int {
        HIPCHK(hipStreamSynchronize(c->stream));
        return MUSE_OK;
    });
    if (rc) return rc;
    c->prof_on = false;
    *count = c->prof_count;
    for (int k = 0; k < c->prof_count && k < cap && ms_out; ++k)
        HIPCHK(hipEventElapsedTime(&ms_out[k], c->prof_ev[2 * k], c->prof_ev[2 * k + 1]));
    return MUSE_OK;
}

int muse_profile_clock_hz(muse_ctx* c, double* hz_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!hz_out) return fail(MUSE_ERR_INVALID, "hz_out is NULL");
    HIPCHK(hipStreamSynchronize(c->stream));
    const unsigned long long* t = c->clock_pin;
    if (t[3] <= t[1] || t[2] <= t[0]) return fail(MUSE_ERR_INVALID, "no profiled launch has run (muse_profile_begin ... end)");
    *hz_out = (double)(t[2] - t[0]) / (double)(t[3] - t[1]) * 1e8;  // s_memrealtime ticks at a constant 100 MHz
    return MUSE_OK;
}

static void base_args(muse_ctx* c, BatchArgs& a, const double* theta) {
    memset(&a, 0, sizeof(a));
    a.N = c->N;
    a.ld = c->ld;
    a.ntheta = c->ntheta;
    for (int k = 0; k <= kMaxTheta; ++k) {
        a.bnd[k] = c->bnd[k];
        a.bnd32[k] = k < nblocks_of(c->ntheta) ? (int)c->bnd[k] : 0x7fffffff;
    }
    make_thetaset(c, theta, a.cur.t);
    a.cur.f_const = theta_const(c, theta);
    if (c->ntheta > 1 && c->model != MUSE_MODEL_NOISE && !kPairModel)  // the big tier (tier_big): every block's coefficients, where its kernels read
        for (int k = 0; k < c->ntheta; ++k) {           // them from the kernarg segment (a launch of several maps overwrites them: maps[])
            a.big.sd[k] = muse_exp(0.5 * theta[k]);
            a.big.iv[k] = muse_exp(-theta[k]);
        }
    a.nmaps = 1;
    a.fid_slot = -1;
    a.nstd = 0x7fffffff;  // no normals-only elements
    set_launch_constants(c, a);
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
    HIPCHK(launch_sample(c->model, a, (uint64_t)sim, dx, dz, dn, c->stream));
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
    HIPCHK(launch_loglike(c->model, a, dx, dz, gdev, c->small_dev, c->stream));
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
    double small[1 + kBigTheta];
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
    double small[1 + kBigTheta];
    HIPCHK(hipMemcpyAsync(small, c->small_dev, sizeof(small), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->ntheta; ++k) g_out[k] = small[1 + k];
    return MUSE_OK;
}

static int enqueue_results_copy(muse_ctx* c, int area, int64_t n) {
    // results are already on their way to pinned host memory; mark the point at which they are complete (the event
    // follows the kernel: its end-of-kernel release has made the result stores visible to the host).  A completion
    // word written by the kernel itself and polled by the host saves the event's ~3 us of idle GPU between two
    // launches (measured: 51.5 vs 54.3 us per 512-sim step), but the results then have to leave with system-scope
    // stores, and with those in flight hipLaunchKernel was measured to block for ~40 us per call: not kept.
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
    a.zhat = dz;  // the element's z lives in the tmp vector (slot 0 of a one-slot view), not in the batch slots
    rc = launch_batch(c, a);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(z_out, dz, (size_t)c->N * sizeof(double), out_kind(mem), c->stream));
    rc = enqueue_results_copy(c, kResultAreas - 1, 1);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = check_error_flag(c);
    if (rc) return rc;
    if (info) *info = c->info_pin[kResultAreas - 1][0];
    return MUSE_OK;
}

// The batched map: `nmaps` independent maps over the same elements, each with a theta of its own (thetas [nmaps][ntheta]), in
// ONE launch.  Scores go to `scores_dev` (any device-accessible buffer; NULL = the area's pinned host block -- muse_comm.cpp
// points it at the send buffer of the RCCL all-gather so that the scores never visit the host between the solver and the
// collective), map m's rows at m * map_stride (>= the element count: a gathered block is padded to rows_per_rank).
struct MapOpts {
    int nmaps = 1;
    int64_t map_stride = 0;   // 0: the element count
    double* scores_dev = nullptr;
    int ncache_mode = 0;      // 1: the batch also stores the normals of its simulations; 2: it loads them (same seed and range
                              // as the storing batch of the same host call).  Silently 0 where the cache does not apply.
    bool lanes_ok = true;                 // the launch may run on the result area's lane (muse_set_concurrency)
};
static int map_async_impl(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                          const double* thetas, double atol, int z0_mode, int area, const MapOpts& o) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!thetas) return fail(MUSE_ERR_INVALID, "theta is NULL");
    if (sim_end < sim_begin || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad sim range");
    if (z0_mode < MUSE_Z0_ZERO || z0_mode > MUSE_Z0_WARM) return fail(MUSE_ERR_INVALID, "bad z0_mode");
    if (area < 0 || area >= kResultAreas) return fail(MUSE_ERR_INVALID, "bad result_area");
    if (o.nmaps < 1 || o.nmaps > kMaxMaps) return fail(MUSE_ERR_INVALID, "nmaps must be in [1, MUSE_MAX_MAPS]");
    if (o.nmaps > 1 && c->ntheta > kMaxTheta)
        return fail(MUSE_ERR_INVALID, "several maps per launch take ntheta <= MUSE_MAX_THETA (the big tier's tables take the place of maps[])");
    if (include_data && !c->has_data) return fail(MUSE_ERR_NODATA, "include_data set but muse_set_data was not called");
    const int64_t n = (sim_end - sim_begin) + (include_data ? 1 : 0);
    const int64_t stride = o.map_stride > 0 ? o.map_stride : n;
    if (stride < n) return fail(MUSE_ERR_INVALID, "map_stride smaller than the element count");
    c->area_failed[area] = false;
    if (n == 0) { c->res_n[area] = 0; c->res_rows[area] = 0; return MUSE_OK; }
    rc = settle_area(c, area);
    if (rc) return rc;
    // the area's lane (muse_set_concurrency).  A lane has resident zhat slots of its own (a streaming solve works in its
    // slot): a warm start, which means "from the MAPs muse_get_zhat / the last plain map left", stays on lane 0; so does a
    // map whose scores feed a device-side collective (its stream order is tied to lane 0's)
    if (c->nlanes > 1 && o.lanes_ok && z0_mode != MUSE_Z0_WARM) {
        rc = use_lane(c, area % c->nlanes);
        if (rc) return rc;
    }
    const int64_t total = n * o.nmaps, rows = stride * o.nmaps;
    if (total > 0x7fffffff || rows > 0x7fffffff) return fail(MUSE_ERR_INVALID, "batch too large");
    rc = ensure_zhat(c, total);
    if (rc) return rc;
    rc = ensure_results(c, area, rows);
    if (rc) return rc;
    BatchArgs a;
    base_args(c, a, thetas);
    a.kind = BATCH_STD;
    a.seed = seed;
    a.atol = atol;
    a.nproblems = (int)total;
    a.include_data = include_data ? 1 : 0;
    a.z0_mode = z0_mode;
    a.store_zhat = 1;
    a.sim_begin = sim_begin;
    a.slot0 = 0;
    a.scores = o.scores_dev ? o.scores_dev : c->scores_dev[area];
    a.info = c->info_dev[area];
    a.nmaps = o.nmaps;
    a.n_per_map = (int)n;
    a.map_stride = stride;
    if (o.nmaps > 1)
        for (int m = 0; m < o.nmaps; ++m) {
            make_thetaset(c, thetas + (size_t)m * c->ntheta, a.maps[m].t);
            a.maps[m].f_const = theta_const(c, thetas + (size_t)m * c->ntheta);
        }
    const int64_t nsim = sim_end - sim_begin;
    if (o.ncache_mode != 0) {   // the caller (muse_run) says which: store with the first iteration, load with the later ones
        if (o.nmaps == 1 && nsim > 0 && o.ncache_mode == 2 && ncache_holds(c, seed, sim_begin, nsim)) {
            a.ncache = c->ncache;
            a.ncache_sim0 = c->nc_sim0;
            a.ncache_count = (int)c->nc_count;
            a.ncache_mode = 2;
        } else if (o.nmaps == 1 && nsim > 0 && o.ncache_mode == 1 && ensure_ncache(c, nsim)) {
            a.ncache = c->ncache;
            a.ncache_sim0 = sim_begin;
            a.ncache_count = (int)nsim;
            a.ncache_mode = 1;
        }
    } else if (o.ncache_mode == 0 && c->nc_auto && o.nmaps == 1 && nsim > 0 && c->cur_lane == 0 && ncache_applies(c)) {
        // A map over simulations the context has drawn before -- every iteration of a muse! loop the HOST drives (muse.py:
        // the same streams at a new theta, src/muse.jl:134,169), a get_J! pass after it: the second time a range is asked
        // for its normals are stored beside the solve, from the third time on they are loaded instead of generated (the
        // same doubles: bit-identical results).  Lane 0 only: stream order is what puts the storing launch before the
        // loading ones.  A map that never repeats (the pipelined cold steps of bench.py) stores nothing.
        const int64_t budget = (int64_t)c->sw.ncache_max_bytes;
        if (ncache_holds(c, seed, sim_begin, nsim)) {
            a.ncache = c->ncache;
            a.ncache_sim0 = c->nc_sim0;
            a.ncache_count = (int)c->nc_count;
            a.ncache_mode = 2;
        } else if (c->nc_seen_count == nsim && c->nc_seen_seed == seed && c->nc_seen_sim0 == sim_begin &&
                   nsim * 2 * c->ld * (int64_t)sizeof(double) <= budget && ensure_ncache(c, nsim)) {
            a.ncache = c->ncache;
            a.ncache_sim0 = sim_begin;
            a.ncache_count = (int)nsim;
            a.ncache_mode = 1;
        }
        c->nc_seen_seed = seed; c->nc_seen_sim0 = sim_begin; c->nc_seen_count = nsim;
    }
    // the area's completion event is signalled by this launch itself; with timing events around the launch (profiling)
    // the plain record after it keeps the order start, kernel, stop, done
    c->launch_done = (c->timing || c->prof_on || c->sw.no_ext_launch) ? nullptr : c->area_done[area];
    c->launch_done_used = false;
    // A storing launch overwrites the cache: whatever it held is gone the moment the launch is issued, and the new range is
    // claimed only once the launch HAS been issued -- a launch that fails leaves no tag behind under which a later map would
    // load slots that were never written (a launch that is issued and then raises the error word: check_error_flag).
    if (a.ncache_mode == 1) c->nc_count = 0;
    rc = launch_batch(c, a);
    c->launch_done = nullptr;
    if (rc) return rc;
    if (a.ncache_mode == 1) { c->nc_seed = seed; c->nc_sim0 = sim_begin; c->nc_count = nsim; }
    c->area_inflight[area] = true;
    c->res_rows[area] = rows;
    if (c->launch_done_used) {
        c->res_n[area] = total;
        return MUSE_OK;
    }
    return enqueue_results_copy(c, area, total);
}
int muse_internal_map_async(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data, int nmaps,
                            const double* thetas, double atol, int z0_mode, int area, int64_t map_stride, double* scores_dev) {
    MapOpts o;
    o.nmaps = nmaps;
    o.map_stride = map_stride;
    o.scores_dev = scores_dev;
    o.lanes_ok = scores_dev == nullptr;   // RCCL transport: the collective stream waits on an event of lane 0's stream
    return map_async_impl(c, seed, sim_begin, sim_end, include_data, thetas, atol, z0_mode, area, o);
}

int muse_map_and_score_batch_async(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                                   const double* theta, double atol, int z0_mode, int area) {
    return map_async_impl(c, seed, sim_begin, sim_end, include_data, theta, atol, z0_mode, area, MapOpts());
}
int muse_map_and_score_multi_async(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data, int nmaps,
                                   const double* thetas, double atol, int z0_mode, int area) {
    MapOpts o;
    o.nmaps = nmaps;
    return map_async_impl(c, seed, sim_begin, sim_end, include_data, thetas, atol, z0_mode, area, o);
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
        MUSE_CPU_RELAX();
    }
    HIPCHK(hipEventSynchronize(ev));
    return MUSE_OK;
}

// the same for everything enqueued on the context's stream
static int muse_wait_event_or_stream(muse_ctx* c) {
    for (int spin = 0; spin < 4000000; ++spin) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return MUSE_OK;
        if (e != hipErrorNotReady) return fail(MUSE_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        MUSE_CPU_RELAX();
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return MUSE_OK;
}

int muse_batch_wait(muse_ctx* c, int area, double* g_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (area < 0 || area >= kResultAreas) return fail(MUSE_ERR_INVALID, "bad result_area");
    rc = muse_wait_event(c->area_done[area]);  // this area only: later launches keep running
    if (rc) return rc;
    c->area_inflight[area] = false;
    rc = check_error_flag(c);
    if (rc) return rc;
    if (c->area_failed[area]) {  // its launch ran beside (or behind) one whose cluster wait expired: its results are not trusted
        c->area_failed[area] = false;
        return fail(MUSE_ERR_HIP, "this result area was in flight when a cluster wait expired inside a solver kernel");
    }
    const int64_t n = c->res_n[area], rows = c->res_rows[area];
    if (g_out && rows) memcpy(g_out, c->scores_pin[area], (size_t)rows * c->ntheta * sizeof(double));
    if (info_out && n) memcpy(info_out, c->info_pin[area], (size_t)n * sizeof(muse_info));
    return MUSE_OK;
}

int muse_map_and_score_batch(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                             const double* theta, double atol, int z0_mode, double* g_out, muse_info* info_out) {
    int rc = muse_map_and_score_batch_async(c, seed, sim_begin, sim_end, include_data, theta, atol, z0_mode, 0);
    if (rc) return rc;
    return muse_batch_wait(c, 0, g_out, info_out);
}

// ---- the muse! outer loop in native code (see muse_hip.h) -------------------------------------------------
static int check_run_args(muse_ctx* c, const double* theta0, const muse_run_options* o, int32_t* niter_out, double* theta_out,
                          double* hist_out, double* gsims_out) {
    if (!theta0 || !o || !niter_out || !theta_out || !hist_out || !gsims_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (c->ntheta > kMaxTheta)
        return fail(MUSE_ERR_INVALID, "the native muse! loops take ntheta <= MUSE_MAX_THETA: run the loop over the batched maps "
                                      "(muse_map_and_score_batch) for more components");
    if (o->nsims < 2 || o->maxsteps < 1) return fail(MUSE_ERR_INVALID, "muse_run needs nsims >= 2 and maxsteps >= 1");
    if (o->prior_kind != 0 && o->prior_kind != 1) return fail(MUSE_ERR_INVALID, "prior_kind must be 0 (flat) or 1 (Gaussian)");
    if (!c->has_data) return fail(MUSE_ERR_NODATA, "muse_run needs the observed data (muse_set_data)");
    return MUSE_OK;
}
static void step_params(const muse_ctx* c, const muse_run_options* o, StepParams& sp) {
    memset(&sp, 0, sizeof sp);
    sp.ntheta = c->ntheta;
    sp.nsims = o->nsims;
    sp.prior_kind = o->prior_kind;
    sp.alpha = o->alpha;
    sp.theta_rtol = o->theta_rtol;
    for (int k = 0; k < c->ntheta; ++k) {
        sp.prior_mean[k] = o->prior_mean[k];
        sp.prior_sigma[k] = o->prior_sigma[k];
    }
}
static int step_error(int err) {
    switch (err) {
        case STEP_SINGULAR_LIKE: return fail(MUSE_ERR_INVALID, "muse_run: singular H^-1_like (zero score variance)");
        case STEP_SINGULAR_POST: return fail(MUSE_ERR_INVALID, "muse_run: singular posterior Hessian");
        case STEP_DOMAIN:
            // sqrt of a negative argument is a DomainError in the reference (an H^-1_post' that is not negative definite)
            return fail(MUSE_ERR_INVALID, "muse_run: DomainError in the convergence test: dtheta' H^-1_post' dtheta > 0 (H^-1_post' is not negative definite)");
        default: return MUSE_OK;
    }
}

// The loop with the algebra on the host: launch, wait, step (step.hpp), launch again.
int muse_run(muse_ctx* c, uint64_t seed, const double* theta0, const muse_run_options* o, int32_t* niter_out,
             double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = check_run_args(c, theta0, o, niter_out, theta_out, hist_out, gsims_out);
    if (rc) return rc;
    const int nt = c->ntheta, S = o->nsims;
    const int64_t H = MUSE_RUN_HIST(nt);
    StepParams sp;
    step_params(c, o, sp);
    StepWork work;
    double theta[kMaxTheta], theta_next[kMaxTheta], mean[kMaxTheta], var[kMaxTheta];
    for (int k = 0; k < nt; ++k) theta[k] = theta0[k];
    std::vector<double> g((size_t)(S + 1) * nt);
    std::vector<muse_info> info((size_t)S + 1);
    int n = 0;
    for (int i = 1; i <= o->maxsteps; ++i) {
        const double t_start = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                   std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-9;
        if (i > 2) {  // convergence on the last two records (src/muse.jl:163-166); a NaN compares false and the loop goes on
            const int cv = step_converged(nt, hist_out + (int64_t)(i - 2) * H, hist_out + (int64_t)(i - 3) * H, o->theta_rtol);
            if (cv < 0) return step_error(STEP_DOMAIN);
            if (cv > 0) break;
        }
        const int z0_mode = (i > 1 || o->z0_warm) ? MUSE_Z0_WARM : MUSE_Z0_ZERO;
        // every iteration re-draws the same streams at a new theta (src/muse.jl:134,169): the first one stores the
        // standard normals, the later ones load them instead of running the generator again
        MapOpts mo;
        mo.ncache_mode = (i == 1 && !ncache_holds(c, seed, 0, S)) ? 1 : 2;
        mo.lanes_ok = false;
        rc = map_async_impl(c, seed, 0, S, 1, theta, o->atol, z0_mode, 0, mo);
        if (rc) return rc;
        rc = muse_batch_wait(c, 0, g.data(), info.data());
        if (rc) return rc;
        double* h = hist_out + (int64_t)(i - 1) * H;
        double* gs = gsims_out + (int64_t)(i - 1) * S * nt;
        memcpy(gs, g.data() + nt, (size_t)S * nt * sizeof(double));
        if (info_out) memcpy(info_out + (int64_t)(i - 1) * (S + 1), info.data(), ((size_t)S + 1) * sizeof(muse_info));
        for (int k = 0; k < nt; ++k) step_moments(k, nt, S, gs, mean[k], var[k]);
        const int err = step_record(sp, theta, g.data(), mean, var, h, theta_next, work);
        if (err != STEP_OK) return step_error(err);
        for (int k = 0; k < nt; ++k) theta[k] = theta_next[k];
        const double t_end = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                 std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-9;
        h[7 * nt + nt * nt] = t_end - t_start;
        n = i;
    }
    *niter_out = n;
    for (int k = 0; k < nt; ++k) theta_out[k] = theta[k];
    return MUSE_OK;
}

// The same loop with NO host in it: ONE launch of the loop kernel (muse_kernels.hip, muse_loop_kernel) runs every
// iteration -- the map, the exchange of the scores between the workgroups (tagged granules, no fence), the step (step.hpp's
// arithmetic, the same bits as the loop above) by every workgroup for itself, the next map.  Every workgroup must be
// resident for that (they meet once per iteration), so the grid is what the occupancy query admits and elements are dealt
// statically, w, w + grid, ...; the cluster placements (an element split, N >= 65 536), whose kernels are not built in loop
// form, and score blocks that do not fit the step's LDS arrays run the host loop -- the same bits either way.
struct RunBuffers {
    unsigned long long* gran = nullptr;   // device [gran_cap] tagged granules of the scores
    int64_t gran_cap = 0;
    unsigned int tag = 0;                 // the last granule tag used (tags grow from run to run: nothing is reset)
    muse_info* info_dummy = nullptr;      // device [info_cap]: where the solver infos go when the caller does not want them
    int64_t info_cap = 0;
    double* scores_dummy = nullptr;       // device [scores_dummy_cap]: the workers' own score block of a SHARDED loop (its stepper
    int64_t scores_dummy_cap = 0;         // writes every rank's scores to the pinned block instead)
    // pinned
    double* hist = nullptr;
    double* scores = nullptr;
    muse_info* infos = nullptr;
    double* theta_out = nullptr;
    int* status = nullptr;
    int64_t cap_hist = 0, cap_scores = 0, cap_infos = 0;
    int grid_place = -1, grid_max = 0;    // the occupancy query's answer for (placement, LDS bytes): asked once
    int grid_scratch = 0;                 // ... and the loop kernel's private segment per lane (bytes of spilled registers)
    size_t grid_lds = 0;
};
// S: simulations of the whole job; nlocal: elements this context solves (S + 1 unless the loop is sharded over ranks)
static int ensure_run_buffers(muse_ctx* c, int maxsteps, int S, int64_t nlocal, bool want_info) {
    if (!c->run) c->run = new RunBuffers();
    RunBuffers& r = *c->run;
    const int nt = c->ntheta;
    if (nlocal * nt > r.scores_dummy_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (r.scores_dummy) HIPCHK(hipFree(r.scores_dummy));
        r.scores_dummy = nullptr; r.scores_dummy_cap = 0;
        HIPCHK(hipMalloc(&r.scores_dummy, (size_t)(nlocal * nt) * sizeof(double)));
        r.scores_dummy_cap = nlocal * nt;
    }
    if (!r.theta_out) {
        HIPCHK(hipHostMalloc(&r.theta_out, kMaxTheta * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&r.status, 64, hipHostMallocDefault));
    }
    const int64_t ngran = (int64_t)2 * nt * (S + 1) + 2 * (kMaxTheta + 1);   // the scores, then the stepper's theta and status
    if (ngran > r.gran_cap || r.tag > 0x70000000u) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (ngran > r.gran_cap) {
            if (r.gran) HIPCHK(hipFree(r.gran));
            r.gran = nullptr; r.gran_cap = 0;
            HIPCHK(hipMalloc(&r.gran, (size_t)ngran * sizeof(unsigned long long)));
            r.gran_cap = ngran;
        }
        HIPCHK(hipMemsetAsync(r.gran, 0, (size_t)r.gran_cap * sizeof(unsigned long long), c->stream));  // tag 0 is never used
        r.tag = 0;
    }
    if (nlocal > r.info_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (r.info_dummy) HIPCHK(hipFree(r.info_dummy));
        r.info_dummy = nullptr; r.info_cap = 0;
        HIPCHK(hipMalloc(&r.info_dummy, (size_t)nlocal * sizeof(muse_info)));
        r.info_cap = nlocal;
    }
    const int64_t nh = (int64_t)maxsteps * MUSE_RUN_HIST(kMaxTheta), ns = (int64_t)maxsteps * (S + 1) * nt,
                  ni = want_info ? (int64_t)maxsteps * nlocal : 0;
    if (nh > r.cap_hist) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (r.hist) HIPCHK(hipHostFree(r.hist));
        r.hist = nullptr; r.cap_hist = 0;
        HIPCHK(hipHostMalloc(&r.hist, (size_t)nh * sizeof(double), hipHostMallocDefault));
        r.cap_hist = nh;
    }
    if (ns > r.cap_scores) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (r.scores) HIPCHK(hipHostFree(r.scores));
        r.scores = nullptr; r.cap_scores = 0;
        HIPCHK(hipHostMalloc(&r.scores, (size_t)ns * sizeof(double), hipHostMallocDefault));
        r.cap_scores = ns;
    }
    if (ni > r.cap_infos) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (r.infos) HIPCHK(hipHostFree(r.infos));
        r.infos = nullptr; r.cap_infos = 0;
        HIPCHK(hipHostMalloc(&r.infos, (size_t)ni * sizeof(muse_info), hipHostMallocDefault));
        r.cap_infos = ni;
    }
    return MUSE_OK;
}
static void free_run_buffers(muse_ctx* c) {
    if (!c->run) return;
    RunBuffers& r = *c->run;
    hipFree(r.gran); hipFree(r.info_dummy); hipFree(r.scores_dummy);
    hipHostFree(r.hist); hipHostFree(r.scores); hipHostFree(r.infos); hipHostFree(r.theta_out); hipHostFree(r.status);
    delete c->run;
    c->run = nullptr;
}

// One rank's share of a loop that is sharded over the ranks of a node (muse_run_sharded, muse_comm.cpp): simulations
// [sim_lo, sim_hi) -- and the data element on the rank that has include_data -- of a job of opt->nsims simulations; every rank's
// workers write their scores to the node's board (pinned host memory mapped by every GPU), every rank's stepper polls all of
// them and takes the same step from the same bits.  null: the whole job on this GPU, scores exchanged through its own memory.
struct LoopShard {
    int64_t sim_lo, sim_hi;
    int include_data;
    unsigned long long* board;   // device pointer of the board: [(nsims + 1) * ntheta][2] granules, the data element's first
    unsigned int tag_base;       // the same on every rank (the board's tags grow from run to run)
    // npeers > 0: a board per GPU in DEVICE memory, each mapped into every rank (hipIpc) -- `board` is this rank's own (what its
    // stepper polls), peers[q] rank q's (this rank's own among them): the workers store their scores into all of them
    unsigned long long* peers[8];
    int npeers;
};
// Can this context run (its share of) the loop as ONE persistent launch?  On success *shape_out / *max_grid_out describe it.
static bool loop_usable(muse_ctx* c, int S, int64_t nlocal, LaunchShape* shape_out, int* max_grid_out) {
    const int nt = c->ntheta;
    const int64_t nprob_total = (int64_t)S + 1;
    const int pl = choose_place(c);
    LaunchShape shape;
    shape.model = c->model; shape.ntheta = nt; shape.place = pl; shape.grid = 0; shape.implicit = false; shape.lds_s = false;
    shape.big = false;  // (the loop kernel runs the resident placements)
    shape.done_event = nullptr;
    const bool xg_lds = pl == P_R512x10;
    shape.lds = place_lds(c, pl) + loop_extra_lds(xg_lds, nprob_total, nt);
    const bool host_only = c->sw.no_loop_kernel;  // tuning aid
    const size_t lds_limit = 160 * 1024;
    // the loop kernel pays where an iteration is short: the resident placements (N <= 10^4).  In the streaming ones an iteration is
    // hundreds of microseconds of HBM traffic, the host's share of it nothing, and the loop kernel's static deal of the elements
    // slower than the map kernel's tickets (N = 30 000 x 512 sims: 377 against 347 us per iteration)
    const bool resident = pl == P_R256x1 || pl == P_R512x4 || pl == P_R512x10;
    if (host_only || c->loop_unfit || !resident || place_is_cluster(pl) || !loop_supported(shape) || shape.lds > lds_limit ||
        (xg_lds && loop_step_bytes(nprob_total, nt) > (size_t)2 * (c->ld + 2) * sizeof(double)))
        return false;
    if (!c->run) c->run = new RunBuffers();
    RunBuffers& r = *c->run;
    if (r.grid_place != pl || r.grid_lds != shape.lds) {
        int mg = 0, scratch = 0;
        if (loop_max_grid(shape, c->num_cus, &mg, &scratch) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        r.grid_place = pl;
        r.grid_lds = shape.lds;
        r.grid_max = mg;
        r.grid_scratch = scratch;
    }
    // A loop kernel most of whose state the compiler spilled (a user's header can do that: the two-parameter model whose score terms
    // multiply every element by a further coefficient -- 630 spilled registers, 420 bytes per lane, against 126 / 120 for the same
    // model with the factor applied to the block's sum) is slower than the host loop, and in that regime the compiler's own spill
    // code was seen to lose uniform values held across the solve (the solver's evaluation counters came back wrong while every
    // number of the run was right: tools/fuzz_user_model.py pairgen, round 6).  Beyond the product's own bound on a kernel's
    // scratch (tools/regs.py, LIBRARY_SCRATCH_LIMIT) the host loop runs.
    constexpr int kLoopScratchLimit = 256;
    if (r.grid_scratch > kLoopScratchLimit && !c->sw.loop_any_scratch) return false;
    // ... and with up to four theta components, or one problem per worker (round 5: the roles as two loops and one copy of the solve
    // in the worker's loop took the loop kernels of one to four components out of scratch, or nearly; those of five to eight still
    // carry 36-44 spilled registers and lose to the host loop when a worker has several problems).  Measured per iteration of a
    // 30-iteration call at 512 sims, host loop / loop kernel (tools/loop_vs_host.py, one box):
    //   N = 10^4: 52.2 / 47.9 us (1 component), 58.0 / 56.9 (2), 71.2 / 69.6 (4), 76.3 / 84.2 (8);  N = 4096: 45.9 / 39.8, 57.4 / 50.8,
    //   60.9 / 53.9, 75.1 / 68.4;  N = 512: 23.6 / 16.2, 26.5 / 20.3, 27.7 / 23.4, 31.6 / 35.3
    // and at 100 sims (one problem per worker) the loop kernel leads for every count (N = 10^4 x 8: 34.5 / 32.0).
    // (round 4, before: 53 / 50, 75 / 79, 89 / 97, 100 / 113 at N = 10^4)
    // Round 6 (tools/runloop_bench.py, 512 sims, five to eight components, host loop / loop kernel per iteration of a 30-iteration call):
    // in the all-register placement of 512 < N <= 4096 (the loop kernel fills the register file: one workgroup per compute unit, two or
    // three problems per worker at 512 sims) 68.2 / 59.4 (N = 2048 x 8), 67.6 / 56.7 (2048 x 5), 70.7 / 63.0 (1000 x 8), 72.7 / 59.6
    // (4096 x 6), and a tie at four problems per worker (N = 4096 x 8, 1000 sims: 101.5 / 103.0): the loop kernel up to three problems
    // per worker there.  N = 10^4 (x and g in LDS, two problems per worker at 512 sims) stays as it was: 77.7 / 76.8 (5), 75.4 / 74.4
    // (6), 77.4 / 80.8 (8).
    const bool any_nt = c->sw.loop_any_ntheta;   // tuning aid / tests: the loop kernel whatever ntheta
    const int64_t per_worker = pl == P_R512x4 ? 3 : 1;
    if (r.grid_max < 2 || (nt > 4 && !any_nt && nlocal > per_worker * (r.grid_max - 1))) return false;
    // ... and with three or four components and MANY problems per worker the map kernel's tickets beat the loop kernel's static deal
    // (N = 10^4, host loop / loop kernel: 4 components 104.6 / 106.6 at 1000 sims, 153.6 / 152.9 at 1500, 192.3 / 211.2 at 2000; 3
    // components 101.7 / 96.6 at 1000, 193.4 / 199.5 at 2000; 1-2 components: the loop kernel at every count measured, to 5000 sims)
    if (pl == P_R512x10 && nt >= 3 && !any_nt && nlocal > 6 * (int64_t)(r.grid_max - 1)) return false;
    if (shape_out) *shape_out = shape;
    if (max_grid_out) *max_grid_out = r.grid_max;
    return true;
}

// The loop as ONE launch.  info_out: [maxsteps][nlocal] (this context's elements).  rc MUSE_LOOP_NOT_RESIDENT (internal, > 0): the
// bounded waits of the kernel expired -- nothing has been reported to the caller yet.
enum { MUSE_LOOP_NOT_RESIDENT = 1001 };
static int run_loop_launch(muse_ctx* c, uint64_t seed, const double* theta0, const muse_run_options* o, const LoopShard* sh,
                           const LaunchShape& shape_in, int max_grid, int32_t* niter_out, double* theta_out, double* hist_out,
                           double* gsims_out, muse_info* info_out) {
    const int nt = c->ntheta, S = o->nsims, maxsteps = o->maxsteps;
    const int64_t H = MUSE_RUN_HIST(nt), nprob_total = (int64_t)S + 1;
    const int64_t sim_lo = sh ? sh->sim_lo : 0, sim_hi = sh ? sh->sim_hi : S;
    const int include_data = sh ? sh->include_data : 1;
    const int64_t nsim_local = sim_hi - sim_lo, nprob = nsim_local + include_data;
    if (nprob < 1) return fail(MUSE_ERR_INVALID, "a rank of a sharded loop needs at least one element");
    LaunchShape shape = shape_in;
    const int pl = shape.place;
    int rc = MUSE_OK;
    if (c->nlanes > 1) {   // (every lane: the loop's workgroups must have the GPU to themselves; lane 0 afterwards)
        rc = muse_synchronize(c);
        if (rc) return rc;
    }
    rc = ensure_run_buffers(c, maxsteps, S, nprob, info_out != nullptr);
    if (rc) return rc;
    RunBuffers& r = *c->run;
    rc = ensure_zhat(c, nprob);
    if (rc) return rc;
    BatchArgs a;
    base_args(c, a, theta0);
    a.kind = BATCH_STD;
    a.seed = seed;
    a.atol = o->atol;
    a.nproblems = (int)nprob;
    a.include_data = include_data;
    a.z0_mode = MUSE_Z0_ZERO;   // (per iteration, set by the kernel)
    a.store_zhat = 1;
    a.sim_begin = sim_lo;
    a.slot0 = 0;
    a.nmaps = 1;
    a.n_per_map = (int)nprob;
    a.map_stride = nprob;
    bool storing = false;
    if (nsim_local > 0) {
        const bool held = ncache_holds(c, seed, sim_lo, nsim_local);   // (an earlier run, or maps of the host driver, drew these streams)
        if (held || ensure_ncache(c, nsim_local)) {
            a.ncache = c->ncache;
            a.ncache_sim0 = held ? c->nc_sim0 : sim_lo;
            a.ncache_count = held ? (int)c->nc_count : (int)nsim_local;
            a.ncache_mode = held ? 2 : 1;   // of the FIRST iteration; the later ones load
            // (a storing run overwrites the cache now and claims its range when it has ENDED well: an aborted loop -- workers
            //  that never ran their first iteration, a bounded wait that expired -- must not leave a tag behind)
            storing = !held;
            if (storing) c->nc_count = 0;
        }
    }
    // where the workers' scores go as granules: this GPU's own buffer, or this rank's rows of the node's board
    const int64_t row0 = sh ? (include_data ? 0 : 1 + sim_lo) : 0;
    a.gran = sh ? sh->board + (size_t)row0 * nt * 2 : r.gran;
    a.gran_sys = sh ? (sh->npeers > 0 ? 2 : 1) : 0;
    if (sh && sh->npeers > 0) {
        a.ngran_peers = sh->npeers;
        for (int q = 0; q < sh->npeers; ++q) a.gran_peers[q] = sh->peers[q] + (size_t)row0 * nt * 2;
    }
    // the common fields, as launch_batch fills them
    a.N = c->N;
    a.ld = c->ld;
    a.ntheta = nt;
    for (int k = 0; k <= kMaxTheta; ++k) {
        a.bnd[k] = c->bnd[k];
        a.bnd32[k] = k < nblocks_of(nt) ? (int)c->bnd[k] : 0x7fffffff;
    }
    a.x_data = c->x_data;
    a.zhat = c->zhat;
    a.work_counter = c->counter;
    a.ticket_base = (int)c->ticket_base;   // (no tickets are drawn: elements are dealt statically)
    a.debug = c->debug & 0xffff;   // (bits 16-21 are host-side: switches.hpp)
    a.stamps = (c->stamps && a.nproblems + 3 <= c->stamps_cap) ? c->stamps : nullptr;   // (+3: the loop kernel's own rows)
    a.csize = 1;
    a.error_flag = c->error_flag;
    // workers (each owns elements w, w + nworkers, ...) and one stepper, all resident at once
    int nworkers = max_grid - 1 < (int)nprob ? max_grid - 1 : (int)nprob;
    if (c->sw.loop_grid >= 1 && c->sw.loop_grid < nworkers) nworkers = c->sw.loop_grid;   // tuning aid (never more than what is resident at once)
    // More elements than workers: the stepper takes elements as well (round 5).  512 simulations + the data on 256 compute units: three
    // of 255 workers had three elements (the others two, the stepper none); now one workgroup has three (the data element and two
    // simulations) and the stepper two like everybody else.  With a worker per element the stepper stays what it was: it polls while
    // the others solve.  tools/runloop_bench.py, wall per iteration of a 30-iteration call: 43.0-43.9 us against 48.1-49.1.
    const bool dedicated = c->sw.loop_dedicated_stepper;   // tuning aid / tests: the layout before
                                                                                            // (muse_debug_flags bit 7 likewise)
    bool solving = !dedicated && !(c->debug & 128) && (int64_t)nworkers < nprob;
    if (oversubscribe(c)) { nworkers = (int)nprob; solving = false; }   // test hook: more workgroups than can be resident at once
    if (solving) ++nworkers;   // (the stepper's slot)
    const int grid = solving ? nworkers : nworkers + 1;
    shape.grid = grid;
    a.scratch_stride = place_scratch_vectors(pl) * c->ld;
    rc = ensure_scratch(c, (size_t)grid * a.scratch_stride);
    if (rc) return rc;
    a.scratch = c->scratch;
    LoopArgs l;
    memset(&l, 0, sizeof l);
    step_params(c, o, l.sp);
    l.maxsteps = maxsteps;
    l.z0_warm = o->z0_warm ? 1 : 0;
    l.hist_out = r.hist;
    l.info_out = info_out ? r.infos : r.info_dummy;
    l.info_stride = info_out ? nprob : 0;
    l.theta_out = r.theta_out;
    l.status = r.status;
    l.nprob_total = (int)nprob_total;
    l.stepper_solves = solving ? 1 : 0;
    {   // workgroup w owns elements first + k * nworkers < nprob, first = w (+ 1 when the data element is the stepper's)
        const int64_t dealt = nprob - ((solving && include_data && (c->debug & 64)) ? 1 : 0);   // (debug bit 6, a tuning aid: the data element is the stepper's)
        l.deal_q = (int)(dealt / nworkers);
        l.deal_r = (int)(dealt % nworkers);
    }
    l.theta_gran = r.gran + (size_t)2 * nt * nprob_total;   // (behind the score granules of an unsharded loop)
    if (sh) {
        l.board = 1;
        l.tag_base = sh->tag_base;
        l.score_gran = sh->board;
        l.scores_out = r.scores_dummy;    // the workers' own block: nowhere (the stepper writes every rank's scores)
        l.scores_stride = 0;
        l.scores_all_out = r.scores;
        // (this GPU's theta granules carry the board's tags: they grow with it)
        if (r.tag > sh->tag_base) HIPCHK(hipMemsetAsync(r.gran + (size_t)2 * nt * nprob_total, 0, (size_t)2 * (kMaxTheta + 1) * sizeof(unsigned long long), c->stream));
        if (r.tag < sh->tag_base + (unsigned)maxsteps + 1) r.tag = sh->tag_base + (unsigned)maxsteps + 1;   // (never backwards: this
                                                                  // GPU's own score granules of unsharded runs keep their last tags)
    } else {
        l.board = 0;
        l.tag_base = r.tag;
        r.tag += (unsigned)maxsteps + 1;
        l.score_gran = r.gran;
        l.scores_out = r.scores;
        l.scores_stride = nprob * nt;
        l.scores_all_out = nullptr;
    }
    r.status[0] = r.status[1] = r.status[2] = 0;
    for (int k = 0; k < nt; ++k) r.theta_out[k] = theta0[k];
    const bool trace = c->sw.run_timing || (c->debug & kDebugRunTiming);   // tuning aid: where a call's own time goes
    auto now_us = [] { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3; };
    const double t_a = trace ? now_us() : 0.0;
    {
        const hipError_t e = launch_loop(shape, a, l, c->stream);
        if (e != hipSuccess) return fail(MUSE_ERR_HIP, std::string("loop kernel launch: ") + hipGetErrorString(e));
    }
    const double t_b = trace ? now_us() : 0.0;
    rc = muse_wait_event_or_stream(c);
    if (rc) return rc;
    if (trace) fprintf(stderr, "[muse_run_device] launch call %.1f us, kernel + wait %.1f us (%d iterations)\n", t_b - t_a, now_us() - t_b, r.status[0]);
    c->area_inflight[0] = false;
    if (r.status[1] == 100) {   // a bounded wait of the loop kernel expired (it also raised the error word: clear it)
        (void)check_error_flag(c);
        return MUSE_LOOP_NOT_RESIDENT;
    }
    rc = check_error_flag(c);
    if (rc) return rc;
    if (r.status[1] != 0) return step_error(r.status[1]);
    const int n = r.status[0];
    // every worker ran its elements' first iteration (the stepper saw all of their scores) and the launch has completed
    if (storing && n >= 1) { c->nc_seed = seed; c->nc_sim0 = sim_lo; c->nc_count = nsim_local; }
    *niter_out = n;
    for (int k = 0; k < nt; ++k) theta_out[k] = r.theta_out[k];
    for (int i = 0; i < n; ++i) {
        memcpy(hist_out + (int64_t)i * H, r.hist + (int64_t)i * H, (size_t)H * sizeof(double));
        memcpy(gsims_out + (int64_t)i * S * nt, r.scores + ((int64_t)i * nprob_total + 1) * nt, (size_t)S * nt * sizeof(double));
    }
    if (info_out) memcpy(info_out, r.infos, (size_t)n * nprob * sizeof(muse_info));
    return MUSE_OK;
}
static int not_resident_error(muse_ctx* c) {
    if (!oversubscribe(c)) c->loop_unfit = true;   // (the test hook provokes the failure on purpose)
    return fail(MUSE_ERR_HIP, "muse_run_device: the workgroups of the loop kernel were not all resident at once (another process on "
                              "this GPU?); muse_run gives the same results with one launch per iteration, and later calls of "
                              "muse_run_device on this context take that loop by themselves");
}

int muse_run_device(muse_ctx* c, uint64_t seed, const double* theta0, const muse_run_options* o, int32_t* niter_out,
                    double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    rc = check_run_args(c, theta0, o, niter_out, theta_out, hist_out, gsims_out);
    if (rc) return rc;
    LaunchShape shape;
    int max_grid = 0;
    // the cluster placements (an element split, N >= 65 536), the streaming ones and several components with several problems per
    // worker run the host loop: the same bits either way
    if (!loop_usable(c, o->nsims, (int64_t)o->nsims + 1, &shape, &max_grid))
        return muse_run(c, seed, theta0, o, niter_out, theta_out, hist_out, gsims_out, info_out);
    rc = run_loop_launch(c, seed, theta0, o, nullptr, shape, max_grid, niter_out, theta_out, hist_out, gsims_out, info_out);
    return rc == MUSE_LOOP_NOT_RESIDENT ? not_resident_error(c) : rc;
}

// ---- for muse_comm.cpp: this rank's share of a sharded loop as ONE launch (muse_run_sharded) -------------------------------------
int muse_internal_loop_usable(muse_ctx* c, int nsims, int64_t nlocal) {
    if (!c || check_ctx(c) != MUSE_OK) return 0;
    return loop_usable(c, nsims, nlocal, nullptr, nullptr) ? 1 : 0;
}
// rc: MUSE_OK; 1001: the kernel's bounded waits expired (workgroups not all resident) -- the caller decides with its peers; < 0: error
int muse_internal_run_loop_shard(muse_ctx* c, uint64_t seed, const double* theta0, const muse_run_options* o, int64_t sim_lo, int64_t sim_hi,
                                 int include_data, void* board_dev, void* const* peer_boards, int npeers, unsigned int tag_base,
                                 int32_t* niter_out, double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !o || !niter_out || !theta_out || !hist_out || !gsims_out || !board_dev) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (c->ntheta > kMaxTheta || o->nsims < 2 || o->maxsteps < 1 || (o->prior_kind != 0 && o->prior_kind != 1) || sim_lo < 0 || sim_hi < sim_lo ||
        sim_hi > o->nsims)
        return fail(MUSE_ERR_INVALID, "bad arguments of a sharded loop");
    if (include_data && !c->has_data) return fail(MUSE_ERR_NODATA, "the rank that holds the data element needs the observed data (muse_set_data)");
    LaunchShape shape;
    int max_grid = 0;
    const int64_t nlocal = (sim_hi - sim_lo) + (include_data ? 1 : 0);
    if (!loop_usable(c, o->nsims, nlocal, &shape, &max_grid)) return fail(MUSE_ERR_INVALID, "the loop kernel cannot run this share");
    LoopShard sh;
    sh.sim_lo = sim_lo; sh.sim_hi = sim_hi; sh.include_data = include_data ? 1 : 0;
    sh.board = (unsigned long long*)board_dev;
    sh.tag_base = tag_base;
    sh.npeers = 0;
    if (peer_boards && npeers > 0 && npeers <= 8) {
        sh.npeers = npeers;
        for (int q = 0; q < npeers; ++q) sh.peers[q] = (unsigned long long*)peer_boards[q];
    }
    rc = run_loop_launch(c, seed, theta0, o, &sh, shape, max_grid, niter_out, theta_out, hist_out, gsims_out, info_out);
    if (rc == MUSE_LOOP_NOT_RESIDENT && !oversubscribe(c)) c->loop_unfit = true;
    return rc;
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

static int ensure_tsample(muse_ctx* c, size_t entries) {
    if (entries <= c->tsample_cap) return MUSE_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->tsample_dev) HIPCHK(hipFree(c->tsample_dev));
    if (c->tsample_pin) HIPCHK(hipHostFree(c->tsample_pin));
    c->tsample_dev = nullptr; c->tsample_pin = nullptr; c->tsample_cap = 0;
    const size_t cap = entries + entries / 2 + 16;
    HIPCHK(hipMalloc(&c->tsample_dev, cap * sizeof(SampleSd)));
    HIPCHK(hipHostMalloc(&c->tsample_pin, cap * sizeof(SampleSd), hipHostMallocDefault));
    c->tsample_cap = cap;
    return MUSE_OK;
}

// The finite-difference map of get_H! (src/muse.jl:426-442) in raw form: for the units (simulation, column) e in
// [e_begin, e_end) of the list (sim_begin, column 0), (sim_begin, column 1), ... -- unit e is column j = e % ntheta of
// simulation sim_begin + e / ntheta -- and G grid points each, the function
//   f(eps) = grad_theta( x(theta0 + eps e_j; the sim's randoms), zhat(x; theta0, start zfid), theta0 )
// at eps = offsets[.][g]: offsets [ntheta][G] shared by the simulations (per_unit false) or [ne][G], one row per unit
// (FiniteDifferences' adaptive step is estimated per call, i.e. per simulation and column).  f_out [ne][G][ntheta],
// info_out [ne][G].  An offset of 0 is allowed (the fiducial theta itself, same randoms).
static int fd_values_impl(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t e_begin, int64_t e_end, const double* theta0,
                          int G, const double* offsets, bool per_unit, double atol, int fid_mode, int64_t fid_sim, double* f_out,
                          muse_info* info_out) {
    const int nt = c->ntheta;
    const int64_t ne = e_end - e_begin;
    if (ne == 0) return MUSE_OK;
    if (G < 1 || G > 64) return fail(MUSE_ERR_INVALID, "grid points per column must be in [1, 64]");
    const int64_t n = ne * G;
    if (n > 0x7fffffff) return fail(MUSE_ERR_INVALID, "batch too large");
    for (int64_t k = 0; k < (per_unit ? ne : nt) * G; ++k)
        if (!isfinite(offsets[k])) return fail(MUSE_ERR_INVALID, "finite-difference offsets must be finite");
    const int64_t s_lo = sim_begin + e_begin / nt, s_hi = sim_begin + (e_end - 1) / nt + 1;  // simulations touched
    const int64_t nsims = s_hi - s_lo;
    // 1. fiducial MAPs at theta0 from zero(z) (src/muse.jl:417-423)
    const int64_t nfid = fid_mode == 0 ? 1 : nsims;
    int rc = settle_area(c, 1);
    if (rc) return rc;
    rc = ensure_zhat(c, nfid);
    if (rc) return rc;
    // every simulation is drawn G*ntheta times (same randoms, perturbed theta; src/muse.jl:426-432): its standard
    // normals are generated once -- by its own fiducial problem (fid_mode 1) or by a normals-only element of the
    // fiducial launch (fid_mode 0) -- and loaded by the perturbed problems
    // (round 5: a cache that already holds these streams -- the muse! loop that ran before get_H!, an earlier call -- is used as it is)
    const bool held = ncache_holds(c, seed, s_lo, nsims);
    const bool cached = held || ensure_ncache(c, nsims);
    const int64_t nc_sim0 = held ? c->nc_sim0 : s_lo, nc_cnt = held ? c->nc_count : nsims;
    const int64_t nprep = nfid + ((cached && !held && fid_mode == 0) ? nsims : 0);
    // Round 6, built, measured and left OFF (MUSE_FD_FOLD / debug flag bit 20 switch it on; tests hold it bit-equal): with the
    // simulations' normals in the cache already the ONE fiducial MAP can be problem 0 of the perturbed problems' launch, whose other
    // problems stage their x while it is solved and wait for its tag before they load their warm start (args.hpp: fd_fold).  It
    // removes a launch, but not the fiducial's ~22 us from the critical path -- every perturbed solve STARTS from that MAP -- and
    // pays for it with an L2 write-back and a 513th problem that makes a third, one-problem round: a 513-problem call (one rank's
    // share of configs[3] on 8 GPUs) 89.7 us against 81.8 us for the two launches, the whole 4097-problem job 279 us either way
    // (gpurun_out/r06d/fd_ab2.log -> profiles/r06_fd_fold_ab.log).
    const bool fold = held && fid_mode == 0 && choose_place(c) == P_R512x10 && (c->sw.fd_fold || (c->debug & kDebugFdFold)) &&
                      n + 1 <= 0x7fffffff;
    rc = ensure_results(c, 1, (n > nprep ? n : nprep) + 1);
    if (rc) return rc;
    // Round 6: the ONE fiducial MAP is the serial part of the call -- every perturbed solve starts from it -- and two thirds of a
    // cold problem is its generator, run by ONE workgroup.  With the perturbed problems' normals in the cache already (so that the
    // fiducial's launch carries nothing else) a kernel of its own draws the fiducial stream's normals with the whole GPU (40
    // workgroups, ~3 us) into a slot of the cache's layout, and the fiducial problem LOADS them: the same function of (seed,
    // stream, element), the same bits.  LDS-resident placement (where the solver can load normals at all).
    const bool fidn = held && !fold && fid_mode == 0 && choose_place(c) == P_R512x10 && !c->sw.no_fid_normals && !(c->debug & kDebugNoFidNormals);
    if (fidn) {
        if (!c->fid_norm && hipMalloc(&c->fid_norm, (size_t)2 * c->ld * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            c->fid_norm = nullptr;
        }
        if (c->fid_norm) HIPCHK(launch_normals(seed, (uint64_t)fid_sim, c->ld, c->fid_norm, c->stream));
    }
    // a sampling entry: exp(theta/2) of every block -- a SampleSd, or kBigTheta doubles in the big tier (solver.hpp, begin)
    const int ts_stride = tier_big(c, choose_place(c), 1) ? kBigTheta : kMaxTheta;
    static_assert(sizeof(SampleSd) == kMaxTheta * sizeof(double) && kBigTheta % kMaxTheta == 0, "sampling entries");
    // a few entries shared by the simulations travel in the launch's own argument block (in the place of maps[]: the launch carries
    // one map) -- no pinned staging, no upload on the stream between the two launches
    const bool ts_in_kernarg = !per_unit && ts_stride == kMaxTheta && (size_t)nt * G * sizeof(SampleSd) <= sizeof(MapTheta) * kMaxMaps;
    if (!ts_in_kernarg) {
        rc = ensure_tsample(c, (size_t)(per_unit ? n : (int64_t)nt * G) * (size_t)(ts_stride / kMaxTheta));
        if (rc) return rc;
    }
    if (!fold) {
        BatchArgs a;
        base_args(c, a, theta0);
        a.kind = BATCH_STD;
        a.seed = seed;
        a.atol = atol;
        a.nproblems = (int)nprep;
        a.nstd = (int)nfid;
        a.norm_sim0 = s_lo;
        if (fidn && c->fid_norm) {   // the launch is the fiducial problem alone: its normals are in fid_norm (slot 0 of a one-stream cache)
            a.ncache = c->fid_norm;
            a.ncache_sim0 = fid_sim;
            a.ncache_count = 1;
            a.ncache_mode = 2;
        } else if (cached) {
            a.ncache = c->ncache;
            a.ncache_sim0 = nc_sim0;
            a.ncache_count = (int)nc_cnt;
            a.ncache_mode = held ? 2 : 1;
            if (!held) c->nc_count = 0;   // (overwritten from here on; claimed below once the whole call has ended well)
        }
        a.include_data = 0;
        a.z0_mode = MUSE_Z0_ZERO;
        a.store_zhat = 1;
        a.sim_begin = fid_mode == 0 ? fid_sim : s_lo;
        a.slot0 = 0;
        a.scores = c->scores_dev[1];
        a.info = c->info_dev[1];
        rc = launch_batch(c, a);
        if (rc) return rc;
    }
    // 2. the perturbed simulations, MAP and score at theta0
    BatchArgs a2;
    base_args(c, a2, theta0);
    {
        std::vector<double> th(nt);
        double* ts_dst = ts_in_kernarg ? reinterpret_cast<double*>(a2.maps) : reinterpret_cast<double*>(c->tsample_pin);
        auto fill = [&](int64_t entry, int j, double off) {
            for (int k = 0; k < nt; ++k) th[k] = theta0[k];
            th[j] = theta0[j] + off;
            double* sd = ts_dst + entry * ts_stride;
#if defined(MUSE_USER_MODEL_HEADER) && defined(MUSE_MODEL_PAIR)
            MapTheta mt;   // the draw's two coefficients of every block at the perturbed theta: the header's, [block][2]
            pair_map_theta(nt, c->bnd, th.data(), mt);
            for (int k = 0; k < ts_stride; ++k) sd[k] = k < nt ? (&mt.t.sd[0])[4 * (k >> 1) + (k & 1)] : 0.0;
#else
            for (int k = 0; k < ts_stride; ++k) sd[k] = k < nt ? muse_exp(0.5 * th[k]) : 0.0;   // (make_map_theta_component's sd)
#endif
        };
        if (per_unit) {
            for (int64_t e = 0; e < ne; ++e)
                for (int g = 0; g < G; ++g) fill(e * G + g, (int)((e_begin + e) % nt), offsets[e * G + g]);
        } else {
            for (int j = 0; j < nt; ++j)
                for (int g = 0; g < G; ++g) fill((int64_t)j * G + g, j, offsets[(int64_t)j * G + g]);
        }
    }
    if (!ts_in_kernarg)
        HIPCHK(hipMemcpyAsync(c->tsample_dev, c->tsample_pin, (size_t)(per_unit ? n : (int64_t)nt * G) * ts_stride * sizeof(double),
                              hipMemcpyHostToDevice, c->stream));
    {
        BatchArgs& a = a2;
        a.kind = BATCH_FD;
        a.seed = seed;
        a.atol = atol;
        a.nproblems = (int)n + (fold ? 1 : 0);
        if (fold) {
            a.fd_fold = 1;
            a.fid_sim = fid_sim;
            a.fid_flag = c->fid_flag;
            a.fid_tag = ++c->fid_tag;
            if (c->fid_tag >= 0x7ffffff0u) {   // (the tag never wraps onto a value the flag may still hold)
                HIPCHK(hipMemsetAsync(c->fid_flag, 0, 64, c->stream));
                c->fid_tag = 0;
                a.fid_tag = ++c->fid_tag;
            }
        }
        a.sim_begin = s_lo;
        a.fd_grid = G;
        a.fd_per_problem = per_unit ? 1 : 0;
        a.p0 = (int)(G * (e_begin - (s_lo - sim_begin) * nt));  // the range may begin inside s_lo's Jacobian
        a.fid_slot = fid_mode == 0 ? 0 : -1;
        a.slot0 = 0;
        a.tsample = ts_in_kernarg ? nullptr : c->tsample_dev;
        a.scores = c->scores_dev[1];
        a.info = c->info_dev[1];
        if (cached) {
            a.ncache = c->ncache;
            a.ncache_sim0 = nc_sim0;
            a.ncache_count = (int)nc_cnt;
            a.ncache_mode = 2;
        }
        rc = launch_batch(c, a);
        if (rc) return rc;
    }
    rc = enqueue_results_copy(c, 1, n);
    if (rc) return rc;
    rc = muse_wait_event_or_stream(c);   // (polls the stream: the runtime's blocking wait wakes up late for a call of tens of microseconds)
    if (rc) return rc;
    rc = check_error_flag(c);
    if (rc) return rc;
    if (cached && !held) { c->nc_seed = seed; c->nc_sim0 = s_lo; c->nc_count = nsims; }   // both launches have completed
    if (f_out) memcpy(f_out, c->scores_pin[1], (size_t)n * nt * sizeof(double));
    if (info_out) memcpy(info_out, c->info_pin[1], (size_t)n * sizeof(muse_info));
    return MUSE_OK;
}

// central_fdm(3,1) with an explicit step: grid (-1, 0, 1), coefficients (-1/2, 0, 1/2) (src/muse.jl:300, src/util.jl:13);
// the centre point has coefficient 0 and is not evaluated.  cols_out [e_end-e_begin][ntheta] (cols[e][i] = d g_i / d theta_j),
// info_out [e_end-e_begin][2] (plus, minus).
static int fd_columns_impl(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t e_begin, int64_t e_end, const double* theta0,
                           const double* step, double atol, int fid_mode, int64_t fid_sim, double* cols_out,
                           muse_info* info_out) {
    const int nt = c->ntheta;
    const int64_t ne = e_end - e_begin;
    if (ne == 0) return MUSE_OK;
    std::vector<double> off((size_t)2 * nt);
    for (int j = 0; j < nt; ++j) {
        if (!(step[j] != 0.0) || !isfinite(step[j])) return fail(MUSE_ERR_INVALID, "step must be finite and non-zero");
        off[2 * j] = step[j];
        off[2 * j + 1] = -step[j];
    }
    int rc = fd_values_impl(c, seed, sim_begin, e_begin, e_end, theta0, 2, off.data(), false, atol, fid_mode, fid_sim, nullptr, info_out);
    if (rc) return rc;
    const double* g = c->scores_pin[1];
    for (int64_t e = 0; e < ne; ++e) {
        const int j = (int)((e_begin + e) % nt);
        const double* gp = g + (e * 2 + 0) * nt;
        const double* gm = g + (e * 2 + 1) * nt;
        for (int i = 0; i < nt; ++i) cols_out[e * nt + i] = (-0.5 * gm[i] + 0.5 * gp[i]) / step[j];
    }
    return MUSE_OK;
}

int muse_fd_values_columns(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t col_begin, int64_t col_end,
                           const double* theta0, int ngrid, const double* offsets, int offsets_per_unit, double atol,
                           int fid_mode, int64_t fid_sim, double* f_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !offsets || !f_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (col_end < col_begin || col_begin < 0 || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad column range");
    if (fid_mode != 0 && fid_mode != 1) return fail(MUSE_ERR_INVALID, "fid_mode must be 0 or 1");
    return fd_values_impl(c, seed, sim_begin, col_begin, col_end, theta0, ngrid, offsets, offsets_per_unit != 0, atol, fid_mode,
                          fid_sim, f_out, info_out);
}

int muse_fd_jacobian_columns(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t col_begin, int64_t col_end,
                             const double* theta0, const double* step, double atol, int fid_mode, int64_t fid_sim,
                             double* cols_out, muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !step || !cols_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (col_end < col_begin || col_begin < 0 || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad column range");
    if (fid_mode != 0 && fid_mode != 1) return fail(MUSE_ERR_INVALID, "fid_mode must be 0 or 1");
    return fd_columns_impl(c, seed, sim_begin, col_begin, col_end, theta0, step, atol, fid_mode, fid_sim, cols_out, info_out);
}

int muse_fd_jacobian_batch(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t sim_end, const double* theta0,
                           const double* step, double atol, int fid_mode, int64_t fid_sim, double* Hs_out,
                           muse_info* info_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !step || !Hs_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (sim_end < sim_begin || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad sim range");
    if (fid_mode != 0 && fid_mode != 1) return fail(MUSE_ERR_INVALID, "fid_mode must be 0 or 1");
    const int nt = c->ntheta;
    const int64_t nsims = sim_end - sim_begin;
    if (nsims == 0) return MUSE_OK;
    std::vector<double> cols((size_t)nsims * nt * nt);
    rc = fd_columns_impl(c, seed, sim_begin, 0, nsims * nt, theta0, step, atol, fid_mode, fid_sim, cols.data(), info_out);
    if (rc) return rc;
    for (int64_t s = 0; s < nsims; ++s)  // the per-sim Jacobian is the hcat of its columns (src/util.jl:25)
        for (int j = 0; j < nt; ++j)
            for (int i = 0; i < nt; ++i) Hs_out[(s * nt + i) * nt + j] = cols[((size_t)(s * nt + j)) * nt + i];
    return MUSE_OK;
}

// Columns [e_begin, e_end) of the same list for the implicit-differentiation H; per_column: one element per column
// (each repeats the simulation's sample and its atol MAP), else one element per simulation (whole simulations only).
static int implicit_impl(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t e_begin, int64_t e_end, bool per_column,
                         const double* theta0, double atol, int cg_maxiter, double* cols_out, int32_t* cg_iters_out) {
    const int nt = c->ntheta;
    const int64_t ne = e_end - e_begin;
    if (!muse_model_has_second())
        return fail(MUSE_ERR_INVALID, "the implicit-differentiation H needs second derivatives, which this model's header does not "
                                      "supply (MUSE_MODEL_SECOND, include/muse_model.h): use the finite-difference entries");
    if (ne == 0) return MUSE_OK;
    if (ne > 0x7fffffff) return fail(MUSE_ERR_INVALID, "batch too large");
    const int64_t s_lo = sim_begin + e_begin / nt, s_hi = sim_begin + (e_end - 1) / nt + 1;
    const int64_t nsims = s_hi - s_lo;
    int rc = settle_area(c, 2);
    if (rc) return rc;
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
    a.imp_split = per_column ? nt : 1;
    a.p0 = per_column ? (int)(e_begin - (s_lo - sim_begin) * nt) : 0;
    a.nproblems = (int)(per_column ? ne : nsims);
    a.sim_begin = s_lo;
    a.slot0 = 0;
    a.scores = c->scores_dev[2];  // [nsims][ntheta][ntheta], H[s][i][j]: only the requested columns are written
    a.info = c->info_dev[2];
    rc = launch_batch(c, a);
    if (rc) return rc;
    rc = enqueue_results_copy(c, 2, nsims * nt);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = check_error_flag(c);
    if (rc) return rc;
    for (int64_t e = 0; e < ne; ++e) {
        const int64_t el = e_begin + e - (s_lo - sim_begin) * nt;  // position in the list that starts at s_lo
        const int64_t s = el / nt;
        const int j = (int)(el % nt);
        for (int i = 0; i < nt; ++i) cols_out[e * nt + i] = c->scores_pin[2][(s * nt + i) * nt + j];
        if (cg_iters_out) cg_iters_out[e] = c->info_pin[2][s * nt + j].iterations;
    }
    return MUSE_OK;
}

int muse_implicit_H_columns(muse_ctx* c, uint64_t seed, int64_t sim_begin, int64_t col_begin, int64_t col_end,
                            const double* theta0, double atol, int cg_maxiter, double* cols_out, int32_t* cg_iters_out) {
    int rc = check_ctx(c);
    if (rc) return rc;
    if (!theta0 || !cols_out) return fail(MUSE_ERR_INVALID, "NULL argument");
    if (col_end < col_begin || col_begin < 0 || sim_begin < 0) return fail(MUSE_ERR_INVALID, "bad column range");
    if (cg_maxiter < 1) return fail(MUSE_ERR_INVALID, "cg_maxiter must be >= 1");
    return implicit_impl(c, seed, sim_begin, col_begin, col_end, true, theta0, atol, cg_maxiter, cols_out, cg_iters_out);
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
    // few simulations and several theta components: one element per (simulation, H column), so that the batch
    // fills the GPU (each element repeats the simulation's sample and its atol MAP, cheap next to ntheta CG solves)
    const int64_t slots = (int64_t)c->num_cus * 2 / (use_cluster(c) ? cluster_size(c) : 1);
    const bool per_column = nt > 1 && nsims * 2 <= slots;
    std::vector<double> cols((size_t)nsims * nt * nt);
    rc = implicit_impl(c, seed, sim_begin, 0, nsims * nt, per_column, theta0, atol, cg_maxiter, cols.data(), cg_iters_out);
    if (rc) return rc;
    for (int64_t s = 0; s < nsims; ++s)
        for (int j = 0; j < nt; ++j)
            for (int i = 0; i < nt; ++i) Hs_out[(s * nt + i) * nt + j] = cols[((size_t)(s * nt + j)) * nt + i];
    return MUSE_OK;
}

}  // extern "C"
