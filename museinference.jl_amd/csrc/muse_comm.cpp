// muse_comm.cpp -- exchange of the per-sim accumulators between ranks (collectives C1-C3 of SURVEY.md §2).
//
// The reference gathers map results to the master process through Distributed.pmap
// (src/util.jl:74-83) and reduces them there (src/muse.jl:183,188,446,529).  Here every rank owns a
// contiguous block of sims on its own GPU and the per-rank score blocks / H accumulators are
// exchanged once per map.  Messages are <= 64 KB, so the cost is latency, not bandwidth.  Two transports:
// RCCL collectives over xGMI (any topology), and -- for the ranks of one node, whose hosts are the consumers
// of the blocks -- a shared-memory segment (shm_gather.hpp) that needs no collective kernel at all.
//
// librccl is opened lazily (dlopen) so that libmuse_hip.so loads on hosts without a usable RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

#include <chrono>
#include <vector>

#include "../../include/muse_hip.h"
#include "shm_gather.hpp"
#include "step.hpp"
#include "switches.hpp"

// Minimal slice of the public RCCL/NCCL C API (rccl.h: ncclGetUniqueId, ncclCommInitRank,
// ncclAllGather, ncclAllReduce, ncclCommDestroy).
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclFloat64 = 8 };
enum { ncclSum = 0 };

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

constexpr int kAreas = 4;  // result areas of the engine (muse_hip.h: result_area in [0, 4))
// One per context/rank: the communicator, a stream of its own for the collectives (so that the
// all-gather of batch k overlaps the solver launch of batch k+1), and per result area the device
// send/receive buffers plus a pinned host landing block.
struct CommState {
    ncclComm_t comm = nullptr;
    int nranks = 1;
    int rank = 0;
    // shared-memory transport (null: RCCL).  Area kAreas of the segment carries the synchronous collectives.
    muse_shm::Gather* shm = nullptr;
    // The node's score board of the sharded DEVICE loop (muse_run_sharded): a region of the shared segment that this process
    // has registered with the HIP runtime, so that its GPU reads and writes it in place.  Every rank's workers store their
    // scores there as tagged granules; every rank's stepper polls all of them.  No host is in the loop between two maps.
    unsigned long long* board_dev = nullptr;   // this GPU's pointer to it (null: not available -- the host loop runs)
    void* board_host = nullptr;
    size_t board_granules = 0;
    unsigned int board_tag = 0;      // the last tag used (the same on every rank: every rank makes the same calls)
    // ... and, where the runtime allows it, a board per GPU in DEVICE memory instead: every rank allocates one and maps every peer's
    // (hipIpcGetMemHandle / hipIpcOpenMemHandle, the handles exchanged through the segment); a worker's score is stored into every
    // rank's board -- posted writes, over xGMI between GPUs -- and a stepper polls its OWN GPU's memory: no PCIe round trip in the
    // iteration.  A collective decision at init (every rank must have opened every handle); the host board otherwise.
    unsigned long long* ipc_own = nullptr;
    void* ipc_peers[8] = {nullptr};   // [rank]: the peer's board as mapped here (own: ipc_own)
    bool ipc_ok = false, ipc_tried = false;
    // the set-up hand-shake of the boards (setup_boards; muse_comm_board_status): 1 every rank saw every peer, 0 some rank did not,
    // -1 not tried; the ranks THIS rank saw; how long its hand-shake kernel polled; what the last muse_run_sharded call ran
    int hs_dev = -1, hs_host = -1;
    unsigned long long hs_mask_dev = 0, hs_mask_host = 0;
    double hs_wait_us[2] = {0.0, 0.0};
    int last_loop = MUSE_BOARD_NONE;
    unsigned int* hs_result = nullptr;   // pinned: the hand-shake kernel's {mask lo, mask hi, ticks}
    const muse::Switches* sw = nullptr;  // the context's environment switches (switches.hpp: read once, by muse_ctx_create)
    bool dev_loop_off = false;       // the device loop failed once on some rank (a shared GPU): host loop from then on, on every rank
    uint64_t seq[kAreas + 1] = {0};   // sequence number of the last exchange per area (the same on every rank)
    size_t nlocal[kAreas] = {0};      // doubles this rank's solver produced PER MAP for the gather in flight
    int nmaps[kAreas] = {1, 1, 1, 1}; // maps of the gather in flight (block per rank: [nmaps][rows_per_rank][ntheta])
    hipStream_t cstream = nullptr;
    bool own_stream = true;
    bool direct_host = false;
    double* send_dev[kAreas] = {nullptr};
    double* recv_dev[kAreas] = {nullptr};
    double* recv_pin[kAreas] = {nullptr};
    size_t cap[kAreas] = {0};      // doubles per rank
    size_t count[kAreas] = {0};    // doubles per rank of the gather in flight
    hipEvent_t kdone[kAreas] = {nullptr};  // solver launch of the area finished (recorded on the solver stream)
    hipEvent_t gdone[kAreas] = {nullptr};  // gathered block landed in recv_pin (recorded on cstream)
    bool pending[kAreas] = {false};
    // The collective of a gathered map is enqueued by a worker thread of the communicator: the caller's thread returns
    // as soon as the solver is launched (measured on the host: solver launch 7.5 us; stream-wait + ncclAllGather +
    // event 14-18 us -- in one thread the N > 1 step was host-bound at ~25 us of enqueueing against a 22 us solver).
    // One communicator, one collective stream: a gather occupies that stream for 35-40 us (one rank), which is what
    // bounds a strongly scaled step; duplicates of the communicator (ncclCommSplit) on streams of their own, one per
    // result area, were measured to make everything worse (collective kernels of several steps resident at once take
    // the CUs the cluster solver needs: 85 us per step against 37).
    std::thread worker;
    std::mutex mu;                   // serialises every use of `comm` (RCCL: one thread at a time) and the worker's error
    std::mutex qmu;                  // the queue
    std::condition_variable cv;
    int queue[kAreas] = {0};         // areas whose gather is to be enqueued, FIFO
    int q_head = 0, q_tail = 0;      // monotonically increasing positions (mod kAreas)
    std::atomic<int> enqueued[kAreas];  // 1: the area's gather has been handed to the collective stream
    bool stop = false;
    int device = 0;
    int worker_rc = 0;               // first error of the worker (reported by muse_batch_wait_gathered)
    std::string worker_err;
    CommState() { for (int a = 0; a < kAreas; ++a) enqueued[a].store(0); }
};

bool load_rccl() {
    if (g_rccl.h) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.AllReduce)
        return false;
    g_rccl.h = h;
    return true;
}
}  // namespace

// accessors implemented in muse_engine.cpp (the context layout is private to that file)
extern "C" {
int muse_ctx_comm_slot(muse_ctx* ctx, void*** comm, int* device, void** stream);
int muse_ctx_comm_buffer(muse_ctx* ctx, size_t doubles, double** buf);
int muse_ctx_area_event(muse_ctx* ctx, int area, void** event, int* ntheta);
int muse_internal_map_async(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data, int nmaps,
                            const double* thetas, double atol, int z0_mode, int area, int64_t map_stride, double* scores_dev);
int muse_ctx_set_comm_reserve(muse_ctx* ctx, int cus);
int muse_ctx_switches(muse_ctx* ctx, const muse::Switches** sw, int* debug);
int muse_set_error(int code, const char* msg);
int muse_wait_event(void* event);
int muse_internal_loop_usable(muse_ctx* ctx, int nsims, int64_t nlocal);
int muse_internal_run_loop_shard(muse_ctx* ctx, uint64_t seed, const double* theta0, const muse_run_options* o, int64_t sim_lo, int64_t sim_hi,
                                 int include_data, void* board_dev, void* const* peer_boards, int npeers, unsigned int tag_base,
                                 int32_t* niter_out, double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out);
}
namespace muse {   // muse_kernels.hip
hipError_t launch_board_handshake(unsigned long long* own, unsigned long long* const* store, int nstore, int nranks, int rank, unsigned int tag,
                                  unsigned long long slot0, unsigned long long ticks, unsigned int* result, hipStream_t st);
}

static CommState* state_of(muse_ctx* ctx, void** stream_out = nullptr) {
    void** slot;
    int device;
    void* stream;
    if (muse_ctx_comm_slot(ctx, &slot, &device, &stream)) return nullptr;
    if (stream_out) *stream_out = stream;
    return (CommState*)*slot;
}

#define RCCLCHK(expr)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != 0)                                                                                      \
            return muse_set_error(MUSE_ERR_RCCL, (std::string(#expr) + ": " +                            \
                                                  (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?")) \
                                                     .c_str());                                           \
    } while (0)
#define HIPCHK2(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return muse_set_error(MUSE_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

// The worker: waits for an area, then  collective stream <- wait(solver done) ; all-gather ; copy ; record(gdone).
static void comm_worker(CommState* st) {
    (void)hipSetDevice(st->device);
    for (;;) {
        int area;
        size_t cnt;
        {
            std::unique_lock<std::mutex> lk(st->qmu);
            st->cv.wait(lk, [&] { return st->stop || st->q_head != st->q_tail; });
            if (st->q_head == st->q_tail) return;  // stop requested and nothing left
            area = st->queue[st->q_head % kAreas];
            st->q_head += 1;
            cnt = st->count[area];
        }
        {
            std::lock_guard<std::mutex> lk(st->mu);  // RCCL calls on one communicator must not overlap
            hipError_t e = hipStreamWaitEvent(st->cstream, st->kdone[area], 0);
            ncclResult_t r = 0;
            if (e == hipSuccess) {
                if (st->direct_host) {
                    // the collective's receive buffer IS the pinned host block (device-mapped): no copy operation follows
                    r = g_rccl.AllGather(st->send_dev[area], st->recv_pin[area], cnt, ncclFloat64, st->comm, st->cstream);
                } else {
                    r = g_rccl.AllGather(st->send_dev[area], st->recv_dev[area], cnt, ncclFloat64, st->comm, st->cstream);
                    if (r == 0)
                        e = hipMemcpyAsync(st->recv_pin[area], st->recv_dev[area], cnt * st->nranks * sizeof(double),
                                           hipMemcpyDeviceToHost, st->cstream);
                }
            }
            if (e == hipSuccess && r == 0) e = hipEventRecord(st->gdone[area], st->cstream);
            if ((e != hipSuccess || r != 0) && st->worker_rc == 0) {
                st->worker_rc = r != 0 ? MUSE_ERR_RCCL : MUSE_ERR_HIP;
                st->worker_err = r != 0 ? std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")
                                        : std::string("collective stream: ") + hipGetErrorString(e);
            }
        }
        st->enqueued[area].store(1, std::memory_order_release);
    }
}

// The id of a shared-memory communicator: magic | capacity per block | segment name.
struct ShmId {
    uint64_t magic;
    uint64_t block_doubles;
    char name[MUSE_UNIQUE_ID_BYTES - 16];
};
static_assert(sizeof(ShmId) == MUSE_UNIQUE_ID_BYTES, "the id travels in the same 128 bytes as RCCL's");
constexpr size_t kShmDefaultBlock = 16384;
constexpr size_t kBoardBytes = 256 * 1024;   // the score board: 32 768 granules -- (nsims + 1) * ntheta <= 16 384
constexpr size_t kBoardHandshakeBytes = 4096;   // behind them: the set-up hand-shake's slots (a granule pair per rank, up to 64 ranks)
constexpr size_t kBoardTotalBytes = kBoardBytes + kBoardHandshakeBytes;
constexpr unsigned long long kBoardHandshakeSlot = kBoardBytes / sizeof(unsigned long long);

#define SHMCHK(st, expr, what)                                                                            \
    do {                                                                                                  \
        const int w_ = (expr);                                                                            \
        if (w_ != 0)                                                                                      \
            return muse_set_error(MUSE_ERR_RCCL, w_ == 2 ? "shared-memory transport: a peer rank failed (" what ")" \
                                                         : "shared-memory transport: timed out waiting for the peers (" what ")"); \
    } while (0)

// Synchronous collectives over the segment's last area, in pieces of at most one block: all-gather into
// recv [nranks][count], or (sum) the sum over ranks, taken in rank order, written back over `send`.
static int shm_allgather(CommState* st, const double* send, size_t count, double* recv, bool sum) {
    muse_shm::Gather& g = *st->shm;
    const size_t B = g.block_doubles;
    for (size_t off = 0; off < count; off += B) {
        const size_t m = count - off < B ? count - off : B;
        const uint64_t s = ++st->seq[kAreas];
        SHMCHK(st, g.wait_consumed(kAreas, s - 1), "collective: previous piece");
        memcpy(g.block(kAreas, st->rank), send + off, m * sizeof(double));
        g.publish_ready(kAreas, s);
        SHMCHK(st, g.wait_ready(kAreas, s), "collective");
        if (sum) {
            for (size_t i = 0; i < m; ++i) {
                double acc = g.block(kAreas, 0)[i];
                for (int q = 1; q < st->nranks; ++q) acc += g.block(kAreas, q)[i];
                recv[off + i] = acc;
            }
        } else {
            for (int q = 0; q < st->nranks; ++q) memcpy(recv + (size_t)q * count + off, g.block(kAreas, q), m * sizeof(double));
        }
        g.publish_consumed(kAreas, s);
    }
    return MUSE_OK;
}

// One hand-shake over a board kind (collective): this rank's one-wavefront kernel stores its tagged pair into `store[0..nstore)` and
// polls `own` for every rank's pair (muse_kernels.hip: board_handshake_kernel), bounded by the switch handshake_ms (default 50 ms);
// then the ranks tell each other whether they saw everybody.  Returns 1 (every rank saw every peer) or 0; mask/wait_us: this rank's.
static int board_handshake(CommState* st, hipStream_t stream, unsigned long long* own, unsigned long long* const* store, int nstore,
                           unsigned int tag, unsigned long long* mask_out, double* wait_us_out) {
    const double bound_ms = st->sw && st->sw->handshake_ms > 0 ? st->sw->handshake_ms : 50.0;
    const unsigned long long ticks = (unsigned long long)(bound_ms * 1e5);   // s_memrealtime: 100 MHz
    bool ok = st->hs_result != nullptr;
    if (ok) {
        st->hs_result[0] = st->hs_result[1] = st->hs_result[2] = 0;
        // (the ranks leave the collective before this within microseconds of each other; the launch itself is a few more)
        ok = muse::launch_board_handshake(own, store, nstore, st->nranks, st->rank, tag, kBoardHandshakeSlot, ticks, st->hs_result, stream) == hipSuccess &&
             hipStreamSynchronize(stream) == hipSuccess;
        (void)hipGetLastError();
    }
    const unsigned long long mask = ok ? ((unsigned long long)st->hs_result[1] << 32) | st->hs_result[0] : 0ull;
    const unsigned long long full = st->nranks >= 64 ? ~0ull : ((1ull << st->nranks) - 1ull);
    *mask_out = mask;
    *wait_us_out = ok ? (double)st->hs_result[2] * 0.01 : 0.0;
    double flag[1] = {(ok && mask == full) ? 1.0 : 0.0};
    if (shm_allgather(st, flag, 1, flag, true) != MUSE_OK) return 0;
    return flag[0] == (double)st->nranks ? 1 : 0;
}

// The score boards of the sharded device loop, set up ONCE per communicator -- by muse_comm_board_status or by the first
// muse_run_sharded call, collective over the segment either way: the one in pinned host memory -- the segment's extra region, mapped
// into this GPU's address space where the runtime allows it -- and a board per GPU in device memory, every rank's mapped into every
// rank (CommState::ipc_*).  Each kind is then PROVED by a hand-shake (board_handshake) before the loop may use it: a store into
// another GPU's board that its poll never sees costs milliseconds here, not a bounded wait inside the first user call.
static void setup_boards(CommState* st, hipStream_t stream) {
    muse_shm::Gather* g = st->shm;
    const muse::Switches none;
    const muse::Switches& sw = st->sw ? *st->sw : none;
    if (hipSetDevice(st->device) == hipSuccess && !st->hs_result) {
        if (hipHostMalloc(&st->hs_result, 64, hipHostMallocDefault) != hipSuccess) st->hs_result = nullptr;
    }
    (void)hipGetLastError();
    if (!sw.no_board && g->extra() && g->extra_bytes() >= kBoardTotalBytes && st->nranks <= 64 &&
        hipHostRegister(g->extra(), g->extra_bytes(), hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, g->extra(), 0) == hipSuccess && dp) {
            st->board_host = g->extra();
            st->board_dev = (unsigned long long*)dp;
            st->board_granules = kBoardBytes / sizeof(unsigned long long);
        } else {
            (void)hipHostUnregister(g->extra());
        }
    }
    (void)hipGetLastError();
    st->ipc_ok = false;
    st->hs_dev = st->hs_host = -1;
    if (st->nranks <= 8) {
        bool ok = !sw.no_ipc_board && !sw.no_board && hipSetDevice(st->device) == hipSuccess;
        hipIpcMemHandle_t mine;
        memset(&mine, 0, sizeof mine);
        if (ok) {
            void* p = nullptr;
            // The board is written by OTHER GPUs and polled by this one: uncached, or fine-grained, device memory.  Plain hipMalloc
            // memory is coarse-grained and cached in this GPU's L2 -- a poll of it is not guaranteed ever to see a peer's store -- so
            // without either kind there is no device board (every rank then uses the host board).
            if (hipExtMallocWithFlags(&p, kBoardTotalBytes, hipDeviceMallocUncached) != hipSuccess) {
                (void)hipGetLastError();
                p = nullptr;
                if (hipExtMallocWithFlags(&p, kBoardTotalBytes, hipDeviceMallocFinegrained) != hipSuccess) {
                    (void)hipGetLastError();
                    p = nullptr;
                }
            }
            ok = p != nullptr && hipMemset(p, 0, kBoardTotalBytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
                 hipIpcGetMemHandle(&mine, p) == hipSuccess;
            st->ipc_own = (unsigned long long*)p;
        }
        (void)hipGetLastError();
        // every rank's {ok, handle}: 1 + 8 doubles per rank (the 64 handle bytes travel as 8 doubles' bit patterns)
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle exchanged as 8 doubles");
        double send[9], recv[9 * 8];
        send[0] = ok ? 1.0 : 0.0;
        memcpy(send + 1, &mine, 64);
        if (shm_allgather(st, send, 9, recv, false) != MUSE_OK) ok = false;
        bool all = ok;
        for (int q = 0; q < st->nranks && all; ++q) all = recv[9 * q] == 1.0;
        if (all) {
            for (int q = 0; q < st->nranks; ++q) {
                if (q == st->rank) { st->ipc_peers[q] = st->ipc_own; continue; }
                hipIpcMemHandle_t h;
                memcpy(&h, recv + 9 * q + 1, 64);
                void* pp = nullptr;
                if (hipIpcOpenMemHandle(&pp, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !pp) { all = false; (void)hipGetLastError(); break; }
                st->ipc_peers[q] = pp;
            }
        }
        double flag[1] = {all ? 1.0 : 0.0};   // did EVERY rank open EVERY handle
        if (shm_allgather(st, flag, 1, flag, true) != MUSE_OK) flag[0] = 0.0;
        st->ipc_ok = flag[0] == (double)st->nranks;
        if (st->ipc_ok) {   // mapped everywhere: now prove that a store into a peer's board is SEEN by the peer's poll
            unsigned long long* stores[8];
            // (test hook: the granules land 64 slots early -- inside the board, beside what the peer polls)
            const long miss = (sw.handshake_fail & 1) ? -64 : 0;
            for (int q = 0; q < st->nranks; ++q) stores[q] = (unsigned long long*)st->ipc_peers[q] + miss;
            st->hs_dev = board_handshake(st, stream, st->ipc_own, stores, st->nranks, 0x7ff00001u, &st->hs_mask_dev, &st->hs_wait_us[0]);
            st->ipc_ok = st->hs_dev == 1;
        }
        if (!st->ipc_ok) {
            for (int q = 0; q < st->nranks; ++q)
                if (q != st->rank && st->ipc_peers[q]) { (void)hipIpcCloseMemHandle(st->ipc_peers[q]); }
            for (int q = 0; q < 8; ++q) st->ipc_peers[q] = nullptr;
            if (st->ipc_own) (void)hipFree(st->ipc_own);
            st->ipc_own = nullptr;
            (void)hipGetLastError();
        }
    }
    {   // the host board: every rank must have it mapped, and every rank's GPU must see every rank's stores through PCIe
        double flag[1] = {st->board_dev ? 1.0 : 0.0};
        if (shm_allgather(st, flag, 1, flag, true) != MUSE_OK) flag[0] = 0.0;
        if (flag[0] == (double)st->nranks) {
            unsigned long long* stores[1] = {st->board_dev + ((sw.handshake_fail & 2) ? -64 : 0)};
            st->hs_host = board_handshake(st, stream, st->board_dev, stores, 1, 0x7ff00002u, &st->hs_mask_host, &st->hs_wait_us[1]);
        }
        if (st->hs_host != 1 && st->board_host) {   // (a board some rank cannot use is no board: the loop is host-driven on every rank)
            (void)hipHostUnregister(st->board_host);
            (void)hipGetLastError();
            st->board_host = nullptr;
            st->board_dev = nullptr;
        }
    }
    if (sw.run_timing)
        fprintf(stderr, "[muse_comm] rank %d of %d: board hand-shake -- device boards %s (saw 0x%llx, %.1f us), host board %s (saw 0x%llx, %.1f us)\n",
                st->rank, st->nranks, st->hs_dev == 1 ? "ok" : st->hs_dev == 0 ? "FAILED" : "not tried", st->hs_mask_dev, st->hs_wait_us[0],
                st->hs_host == 1 ? "ok" : st->hs_host == 0 ? "FAILED" : "not tried", st->hs_mask_host, st->hs_wait_us[1]);
}

extern "C" {

int muse_comm_unique_id(void* id_out) {
    if (!id_out) return muse_set_error(MUSE_ERR_INVALID, "id_out is NULL");
    if (!load_rccl()) return muse_set_error(MUSE_ERR_RCCL, "librccl could not be loaded");
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, MUSE_UNIQUE_ID_BYTES);
    return MUSE_OK;
}

int muse_comm_unique_id_ex(int transport, int64_t block_doubles, void* id_out) {
    if (transport == MUSE_TRANSPORT_RCCL) return muse_comm_unique_id(id_out);
    if (transport != MUSE_TRANSPORT_SHM) return muse_set_error(MUSE_ERR_INVALID, "unknown transport");
    if (!id_out || block_doubles < 0) return muse_set_error(MUSE_ERR_INVALID, "bad arguments");
    static std::atomic<unsigned> counter{0};
    ShmId id;
    memset(&id, 0, sizeof id);
    id.magic = muse_shm::kMagic;
    id.block_doubles = ((block_doubles ? (size_t)block_doubles : kShmDefaultBlock) + 7) & ~(size_t)7;  // whole cache lines
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id.name, sizeof id.name, "/muse_gather_%d_%llx_%u", (int)getpid(),
             (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec, counter.fetch_add(1));
    memcpy(id_out, &id, MUSE_UNIQUE_ID_BYTES);
    return MUSE_OK;
}

int muse_comm_transport(muse_ctx* ctx, int* transport_out) {
    CommState* st = state_of(ctx);
    if (!st) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    if (!transport_out) return muse_set_error(MUSE_ERR_INVALID, "transport_out is NULL");
    *transport_out = st->shm ? MUSE_TRANSPORT_SHM : MUSE_TRANSPORT_RCCL;
    return MUSE_OK;
}

int muse_comm_ranks_seen(muse_ctx* ctx, int* nranks_out) {
    CommState* st = state_of(ctx);
    if (!st) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    if (!nranks_out) return muse_set_error(MUSE_ERR_INVALID, "nranks_out is NULL");
    if (st->shm) {
        *nranks_out = st->shm->attached();
        return MUSE_OK;
    }
    if (!g_rccl.CommCount) return muse_set_error(MUSE_ERR_RCCL, "librccl has no ncclCommCount");
    std::lock_guard<std::mutex> lk(st->mu);
    RCCLCHK(g_rccl.CommCount(st->comm, nranks_out));
    return MUSE_OK;
}

int muse_comm_board_status(muse_ctx* ctx, int status_out[6], double wait_us_out[2]) {
    void* stream = nullptr;
    CommState* st = state_of(ctx, &stream);
    if (!st) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    if (!status_out || !wait_us_out) return muse_set_error(MUSE_ERR_INVALID, "NULL argument");
    if (st->shm && !st->ipc_tried) {   // collective: every rank is here (or in its first muse_run_sharded call)
        st->ipc_tried = true;
        setup_boards(st, (hipStream_t)stream);
    }
    status_out[0] = st->ipc_ok ? MUSE_BOARD_DEVICE : st->board_dev ? MUSE_BOARD_HOST : MUSE_BOARD_NONE;
    status_out[1] = st->hs_dev;
    status_out[2] = st->hs_host;
    status_out[3] = (int)(st->hs_mask_dev & 0x7fffffffull);
    status_out[4] = (int)(st->hs_mask_host & 0x7fffffffull);
    status_out[5] = st->last_loop;
    wait_us_out[0] = st->hs_wait_us[0];
    wait_us_out[1] = st->hs_wait_us[1];
    return MUSE_OK;
}

int muse_comm_init(muse_ctx* ctx, int nranks, int rank, const void* id) {
    void** slot;
    int device;
    void* stream;
    int rc = muse_ctx_comm_slot(ctx, &slot, &device, &stream);
    if (rc) return rc;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return muse_set_error(MUSE_ERR_INVALID, "bad communicator arguments");
    if (*slot) return muse_set_error(MUSE_ERR_INVALID, "communicator already initialised");
    {
        ShmId sid;
        memcpy(&sid, id, sizeof sid);
        if (sid.magic == muse_shm::kMagic) {
            sid.name[sizeof sid.name - 1] = 0;
            muse_shm::Gather* g = new muse_shm::Gather();
            const muse::Switches* sw = nullptr;
            (void)muse_ctx_switches(ctx, &sw, nullptr);
            if (sw && sw->shm_timeout_s > 0) g->timeout_s = sw->shm_timeout_s;
            std::string err;
            if (!g->open(sid.name, nranks, rank, kAreas + 1, (size_t)sid.block_doubles, err, kBoardTotalBytes)) {
                delete g;
                return muse_set_error(MUSE_ERR_RCCL, ("shared-memory transport: " + err).c_str());
            }
            CommState* st = new CommState();
            st->shm = g;
            st->nranks = nranks;
            st->rank = rank;
            st->device = device;
            st->sw = sw;
            *slot = st;
            return MUSE_OK;   // (the score boards are set up by muse_comm_board_status or the first muse_run_sharded call: collective, and
                              //  nothing a communicator that only gathers maps has to go through)
        }
    }
    if (!load_rccl()) return muse_set_error(MUSE_ERR_RCCL, "librccl could not be loaded");
    HIPCHK2(hipSetDevice(device));
    ncclUniqueId uid;
    memcpy(&uid, id, MUSE_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&comm, nranks, uid, rank));
    CommState* st = new CommState();
    st->comm = comm;
    st->nranks = nranks;
    st->rank = rank;
    (void)muse_ctx_switches(ctx, &st->sw, nullptr);
    if (st->sw && st->sw->comm_one_stream) st->cstream = (hipStream_t)stream;  // tuning aid: collectives in line with the solver
    else {
        // Highest priority: the persistent solver kernel fills every CU (LDS- and VGPR-bound, nothing can
        // co-reside), so a collective can only be dispatched in the gap between two solver launches -- with
        // default priority it loses that race to the next solver launch (measured: 86 vs 73 us per step).
        int lo = 0, hi = 0;
        HIPCHK2(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK2(hipStreamCreateWithPriority(&st->cstream, hipStreamNonBlocking, hi));
    }
    st->direct_host = st->sw && st->sw->comm_direct_host;
    st->own_stream = st->cstream != (hipStream_t)stream;
    for (int a = 0; a < kAreas; ++a) {
        HIPCHK2(hipEventCreateWithFlags(&st->kdone[a], hipEventDisableTiming));
        HIPCHK2(hipEventCreateWithFlags(&st->gdone[a], hipEventDisableTiming));
    }
    st->device = device;
    st->worker = std::thread(comm_worker, st);
    *slot = st;
    muse_ctx_set_comm_reserve(ctx, 16);  // the all-gather kernel of step k runs beside the (cluster) solver launch of step k+1
    return MUSE_OK;
}

int muse_comm_destroy(muse_ctx* ctx) {
    void** slot;
    int device;
    void* stream;
    int rc = muse_ctx_comm_slot(ctx, &slot, &device, &stream);
    if (rc) return rc;
    if (CommState* st = (CommState*)*slot) {
        if (st->shm) {
            if (st->board_host || st->ipc_own || st->hs_result) {
                hipSetDevice(device);
                (void)hipDeviceSynchronize();   // (nothing of this process may still be polling the board)
            }
            if (st->board_host) (void)hipHostUnregister(st->board_host);
            if (st->hs_result) (void)hipHostFree(st->hs_result);
            if (st->ipc_ok) {
                // (every rank closes its views before anyone frees: one more exchange; a peer that is gone already just times out)
                for (int q = 0; q < st->nranks; ++q)
                    if (q != st->rank && st->ipc_peers[q]) (void)hipIpcCloseMemHandle(st->ipc_peers[q]);
                double f[1] = {1.0};
                const double keep = st->shm->timeout_s;
                st->shm->timeout_s = keep < 5.0 ? keep : 5.0;
                (void)shm_allgather(st, f, 1, f, true);
                st->shm->timeout_s = keep;
                (void)hipFree(st->ipc_own);
                (void)hipGetLastError();
            }
            delete st->shm;
            delete st;
            *slot = nullptr;
            return MUSE_OK;
        }
        hipSetDevice(device);
        muse_ctx_set_comm_reserve(ctx, 0);
        {
            std::lock_guard<std::mutex> lk(st->qmu);
            st->stop = true;
        }
        st->cv.notify_all();
        if (st->worker.joinable()) st->worker.join();
        if (st->cstream) hipStreamSynchronize(st->cstream);
        if (st->comm && g_rccl.h) g_rccl.CommDestroy(st->comm);
        for (int a = 0; a < kAreas; ++a) {
            hipFree(st->send_dev[a]);
            hipFree(st->recv_dev[a]);
            hipHostFree(st->recv_pin[a]);
            if (st->kdone[a]) hipEventDestroy(st->kdone[a]);
            if (st->gdone[a]) hipEventDestroy(st->gdone[a]);
        }
        if (st->cstream && st->own_stream) hipStreamDestroy(st->cstream);
        delete st;
    }
    *slot = nullptr;
    return MUSE_OK;
}

int muse_allgather_scores(muse_ctx* ctx, const double* send, int64_t count, double* recv_out) {
    void** slot;
    int device, nranks = 0;
    void* stream;
    int rc = muse_ctx_comm_slot(ctx, &slot, &device, &stream);
    if (rc) return rc;
    if (!*slot) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    if (!send || !recv_out || count < 0) return muse_set_error(MUSE_ERR_INVALID, "bad arguments");
    if (count == 0) return MUSE_OK;
    if (((CommState*)*slot)->shm) return shm_allgather((CommState*)*slot, send, (size_t)count, recv_out, false);
    ncclComm_t comm = ((CommState*)*slot)->comm;
    nranks = ((CommState*)*slot)->nranks;
    double* buf;
    rc = muse_ctx_comm_buffer(ctx, (size_t)count * (size_t)(nranks + 1), &buf);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(((CommState*)*slot)->mu);
    HIPCHK2(hipMemcpyAsync(buf, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
    RCCLCHK(g_rccl.AllGather(buf, buf + count, (size_t)count, ncclFloat64, comm, st));
    HIPCHK2(hipMemcpyAsync(recv_out, buf + count, (size_t)count * nranks * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK2(hipStreamSynchronize(st));
    return MUSE_OK;
}

int muse_allreduce_sum(muse_ctx* ctx, double* hostbuf, int64_t count) {
    void** slot;
    int device;
    void* stream;
    int rc = muse_ctx_comm_slot(ctx, &slot, &device, &stream);
    if (rc) return rc;
    if (!*slot) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    if (!hostbuf || count < 0) return muse_set_error(MUSE_ERR_INVALID, "bad arguments");
    if (count == 0) return MUSE_OK;
    if (((CommState*)*slot)->shm) return shm_allgather((CommState*)*slot, hostbuf, (size_t)count, hostbuf, true);
    double* buf;
    rc = muse_ctx_comm_buffer(ctx, (size_t)count, &buf);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(((CommState*)*slot)->mu);
    HIPCHK2(hipMemcpyAsync(buf, hostbuf, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
    RCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, ((CommState*)*slot)->comm, st));
    HIPCHK2(hipMemcpyAsync(hostbuf, buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK2(hipStreamSynchronize(st));
    return MUSE_OK;
}

// ---- sharded map: solver launch + device-side all-gather, pipelined over the result areas ------------
static int ensure_gather_buffers(CommState* st, int area, size_t doubles_per_rank) {
    if (doubles_per_rank <= st->cap[area]) return MUSE_OK;
    HIPCHK2(hipStreamSynchronize(st->cstream));
    hipFree(st->send_dev[area]);
    hipFree(st->recv_dev[area]);
    hipHostFree(st->recv_pin[area]);
    st->send_dev[area] = st->recv_dev[area] = st->recv_pin[area] = nullptr;
    st->cap[area] = 0;
    const size_t cap = doubles_per_rank + doubles_per_rank / 2 + 16;
    if (hipMalloc(&st->send_dev[area], cap * sizeof(double)) != hipSuccess ||
        hipMalloc(&st->recv_dev[area], cap * st->nranks * sizeof(double)) != hipSuccess)
        return muse_set_error(MUSE_ERR_ALLOC, "hipMalloc(gather buffers) failed");
    HIPCHK2(hipHostMalloc(&st->recv_pin[area], cap * st->nranks * sizeof(double), hipHostMallocDefault));
    st->cap[area] = cap;
    return MUSE_OK;
}

int muse_map_and_score_multi_gather_async(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end,
                                          int include_data, int nmaps, const double* thetas, double atol, int z0_mode,
                                          int64_t rows_per_rank, int area) {
    void* stream = nullptr;
    CommState* st = state_of(ctx, &stream);
    if (!st) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    void* area_ev = nullptr;
    int nt = 0;
    int rc = muse_ctx_area_event(ctx, area, &area_ev, &nt);
    if (rc) return rc;
    const int64_t n = (sim_end - sim_begin) + (include_data ? 1 : 0);
    if (sim_end < sim_begin || rows_per_rank < n || rows_per_rank < 1)
        return muse_set_error(MUSE_ERR_INVALID, "rows_per_rank must be >= this rank's element count (and >= 1)");
    if (nmaps < 1) return muse_set_error(MUSE_ERR_INVALID, "nmaps must be >= 1");
    const size_t cnt = (size_t)nmaps * (size_t)rows_per_rank * (size_t)nt;   // block per rank: [nmaps][rows_per_rank][ntheta]
    if (st->pending[area]) return muse_set_error(MUSE_ERR_INVALID, "a gather is still in flight on this result area");
    if (st->shm) {
        // the plain launch (scores to this area's pinned block); the exchange happens in muse_batch_wait_gathered
        if (cnt > st->shm->block_doubles)
            return muse_set_error(MUSE_ERR_INVALID, "nmaps * rows_per_rank * ntheta exceeds the block capacity the communicator's id was "
                                                    "created with (muse_comm_unique_id_ex: block_doubles)");
        rc = muse_internal_map_async(ctx, seed, sim_begin, sim_end, include_data, nmaps, thetas, atol, z0_mode, area, rows_per_rank,
                                     nullptr);
        if (rc) return rc;
        st->count[area] = cnt;
        st->nlocal[area] = (size_t)n * nt;
        st->nmaps[area] = nmaps;
        st->seq[area] += 1;
        st->pending[area] = true;
        return MUSE_OK;
    }
    rc = ensure_gather_buffers(st, area, cnt);
    if (rc) return rc;
    hipStream_t ks = (hipStream_t)stream;
    // (the area's previous gather has been awaited -- pending is clear -- so its send buffer is free again)
    if (n < rows_per_rank)  // padding rows of a short block are zeros
        HIPCHK2(hipMemsetAsync(st->send_dev[area], 0, cnt * sizeof(double), ks));
    rc = muse_internal_map_async(ctx, seed, sim_begin, sim_end, include_data, nmaps, thetas, atol, z0_mode, area, rows_per_rank,
                                 st->send_dev[area]);
    if (rc) return rc;
    HIPCHK2(hipEventRecord(st->kdone[area], ks));
    st->count[area] = cnt;
    st->pending[area] = true;
    st->enqueued[area].store(0, std::memory_order_relaxed);
    {   // hand the collective to the worker
        std::lock_guard<std::mutex> lk(st->qmu);
        st->queue[st->q_tail % kAreas] = area;
        st->q_tail += 1;
    }
    st->cv.notify_one();
    return MUSE_OK;
}

int muse_map_and_score_batch_gather_async(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end,
                                          int include_data, const double* theta, double atol, int z0_mode,
                                          int64_t rows_per_rank, int area) {
    return muse_map_and_score_multi_gather_async(ctx, seed, sim_begin, sim_end, include_data, 1, theta, atol, z0_mode,
                                                 rows_per_rank, area);
}

int muse_batch_wait_gathered(muse_ctx* ctx, int area, double* g_all_out, muse_info* info_out) {
    CommState* st = state_of(ctx);
    if (!st) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    if (area < 0 || area >= kAreas) return muse_set_error(MUSE_ERR_INVALID, "bad result_area");
    if (!st->pending[area]) return muse_set_error(MUSE_ERR_INVALID, "no gather in flight on this result area");
    if (st->shm) {
        muse_shm::Gather& g = *st->shm;
        const uint64_t s = st->seq[area];
        const size_t cnt = st->count[area];
        st->pending[area] = false;
        // every rank has copied this rank's previous block of the area out (true at once in a pipelined loop)
        SHMCHK(st, g.wait_consumed(area, s - 1), "gathered map: previous block");
        double* mine = g.block(area, st->rank);
        int rc = muse_batch_wait(ctx, area, mine, info_out);  // solver's completion, error flag; scores -> my block
        if (rc) {
            g.raise_abort();  // the peers must not wait a minute for a block that will not come
            return rc;
        }
        {   // padding rows of a short block (after every map's own rows) are zeros
            const size_t per_map = cnt / (size_t)st->nmaps[area];
            if (st->nlocal[area] < per_map)
                for (int m = 0; m < st->nmaps[area]; ++m)
                    memset(mine + (size_t)m * per_map + st->nlocal[area], 0, (per_map - st->nlocal[area]) * sizeof(double));
        }
        g.publish_ready(area, s);
        SHMCHK(st, g.wait_ready(area, s), "gathered map");
        if (g_all_out)
            for (int q = 0; q < st->nranks; ++q) memcpy(g_all_out + (size_t)q * cnt, g.block(area, q), cnt * sizeof(double));
        g.publish_consumed(area, s);
        return MUSE_OK;
    }
    {   // the worker is microseconds behind; bounded all the same (a worker that has died must not hang the caller)
        const double t0 = muse_shm::now_s();
        unsigned spins = 0;
        while (!st->enqueued[area].load(std::memory_order_acquire)) {
            MUSE_CPU_RELAX();
            if ((++spins & 0xfff) == 0 && muse_shm::now_s() - t0 > 30.0) {
                st->pending[area] = false;
                return muse_set_error(MUSE_ERR_RCCL, "the communicator's worker thread did not enqueue the gather within 30 s");
            }
        }
    }
    st->pending[area] = false;
    {
        std::lock_guard<std::mutex> lk(st->mu);
        if (st->worker_rc) {
            const int wrc = st->worker_rc;
            st->worker_rc = 0;
            return muse_set_error(wrc, st->worker_err.c_str());
        }
    }
    int rc = muse_wait_event(st->gdone[area]);
    if (rc) return rc;
    if (g_all_out) memcpy(g_all_out, st->recv_pin[area], st->count[area] * st->nranks * sizeof(double));
    return muse_batch_wait(ctx, area, nullptr, info_out);  // the solver's own completion, error flag, local infos
}

}  // extern "C"

// ---- the muse! outer loop over the ranks of a communicator (src/muse.jl:159-232 with the pmap of :169 over a pool of GPUs) ---
// muse_run with this rank's share of every map: rank r owns the contiguous block of simulations block_partition gives it
// (distributed.py: the first nsims mod nranks ranks get one more), the data element lives on rank 0; per iteration ONE
// gathered map (muse_map_and_score_batch_gather_async: the solver launch and the exchange of the score blocks), after which
// every rank holds every score in simulation order and takes the same step (step.hpp) -- so the ranks agree on theta bit for
// bit without exchanging it, and the trajectory is the unsharded muse_run's.  Nothing but the loop is new: no Python, no
// torch tensor and no allocation sits between two maps.  info_out (may be NULL): THIS rank's solver infos,
// [maxsteps][count of this rank's elements] (the data element first on rank 0).
extern "C" int muse_run_sharded(muse_ctx* ctx, uint64_t seed, const double* theta0, const muse_run_options* o, int32_t* niter_out,
                                double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out) {
    using namespace muse;
    CommState* st = state_of(ctx);
    if (!st) return muse_set_error(MUSE_ERR_INVALID, "muse_comm_init was not called");
    if (!theta0 || !o || !niter_out || !theta_out || !hist_out || !gsims_out) return muse_set_error(MUSE_ERR_INVALID, "NULL argument");
    if (o->nsims < 2 || o->maxsteps < 1) return muse_set_error(MUSE_ERR_INVALID, "muse_run_sharded needs nsims >= 2 and maxsteps >= 1");
    if (o->prior_kind != 0 && o->prior_kind != 1) return muse_set_error(MUSE_ERR_INVALID, "prior_kind must be 0 (flat) or 1 (Gaussian)");
    void* ev = nullptr;
    int nt = 0;
    int rc = muse_ctx_area_event(ctx, 0, &ev, &nt);
    if (rc) return rc;
    if (nt > kMaxTheta) return muse_set_error(MUSE_ERR_INVALID, "the native muse! loops take ntheta <= MUSE_MAX_THETA");
    const int S = o->nsims, world = st->nranks, rank = st->rank;
    const int64_t H = MUSE_RUN_HIST(nt);
    auto block = [&](int r, int64_t& lo, int64_t& hi) {
        const int64_t base = S / world, extra = S % world;
        lo = (int64_t)r * base + (r < extra ? r : extra);
        hi = lo + base + (r < extra ? 1 : 0);
    };
    int64_t rows = 0, lo = 0, hi = 0;
    for (int r = 0; r < world; ++r) {
        int64_t l, h;
        block(r, l, h);
        const int64_t cnt = (h - l) + (r == 0 ? 1 : 0);
        rows = cnt > rows ? cnt : rows;
    }
    block(rank, lo, hi);
    const int64_t nlocal = (hi - lo) + (rank == 0 ? 1 : 0);
    // ---- the device loop: ONE persistent launch per rank runs every iteration; the ranks' scores meet on the node's board (pinned host
    // memory that every GPU maps), every rank's stepper takes the same step from the same bits -- no host between two maps.  Every
    // rank must take the same loop: the decision is the minimum over the ranks of what each can do.
    if (st->shm) {
        void* lane0 = nullptr;
        (void)state_of(ctx, &lane0);
        if (!st->ipc_tried) {   // (every rank makes its first call here together: collective; failure on any rank = the next board on all)
            st->ipc_tried = true;
            setup_boards(st, (hipStream_t)lane0);
        }
        int dbg = 0;
        (void)muse_ctx_switches(ctx, nullptr, &dbg);
        const bool sw_host_board = (st->sw && st->sw->host_board) || (dbg & muse::kDebugHostBoard);
        const bool sw_host_loop = (st->sw && st->sw->sharded_host_loop) || (dbg & muse::kDebugShardedHostLoop);
        const bool ipc = st->ipc_ok && !sw_host_board;   // (the same answer on every rank: every rank sets the same switches)
        // (nlocal >= 1: with fewer simulations than ranks some rank owns no element -- its loop launch would be refused while its
        //  peers' ran: the minimum over the ranks sends such a job to the host-driven loop)
        const bool want = (ipc || st->board_dev) && !st->dev_loop_off && !sw_host_loop && nlocal >= 1 &&
                          (uint64_t)(S + 1) * (uint64_t)nt * 2 <= kBoardBytes / sizeof(unsigned long long) && st->board_tag < 0x70000000u &&
                          muse_internal_loop_usable(ctx, S, nlocal) != 0;
        double flag[1] = {want ? 1.0 : 0.0};
        rc = shm_allgather(st, flag, 1, flag, true);
        if (rc) return rc;
        st->last_loop = flag[0] != (double)world ? MUSE_BOARD_NONE : ipc ? MUSE_BOARD_DEVICE : MUSE_BOARD_HOST;
        if ((st->sw && st->sw->run_timing) || (dbg & muse::kDebugRunTiming))   // tuning aid / tests: which loop, through which board
            fprintf(stderr, "[muse_run_sharded] rank %d of %d: %s\n", rank, world,
                    flag[0] != (double)world ? "host-driven loop" : ipc ? "persistent launch, boards in device memory (hipIpc)"
                                                                        : "persistent launch, board in pinned host memory");
        if (flag[0] == (double)world) {
            const unsigned int tag_base = st->board_tag;
            st->board_tag += (unsigned)o->maxsteps + 1;
            rc = muse_internal_run_loop_shard(ctx, seed, theta0, o, lo, hi, rank == 0 ? 1 : 0, ipc ? (void*)st->ipc_own : (void*)st->board_dev,
                                              ipc ? st->ipc_peers : nullptr, ipc ? world : 0, tag_base, niter_out, theta_out, hist_out,
                                              gsims_out, info_out);
            // a rank whose workgroups were not all resident (rc 1001) stalls every rank's stepper: all of them time out -- but
            // make the outcome a collective decision anyway; and a rank that FAILED (rc < 0: before or after its launch) takes every
            // rank out with an error -- its peers' steppers have waited for scores that never came
            double bad[1] = {(rc == 1001 ? 1.0 : 0.0) + (rc < 0 ? 1000.0 : 0.0)};
            const int rc2 = shm_allgather(st, bad, 1, bad, true);
            if (rc < 0) return rc;
            if (rc2) return rc2;
            if (bad[0] >= 1000.0) {
                st->last_loop = MUSE_BOARD_NONE;
                return muse_set_error(MUSE_ERR_RCCL, "muse_run_sharded: a peer rank's share of the persistent loop failed (its own call reports why)");
            }
            if (bad[0] == 0.0) return MUSE_OK;
            st->last_loop = MUSE_BOARD_NONE;
            st->dev_loop_off = true;
            if (o->z0_warm)   // (the aborted attempt has touched the resident MAPs the run was to start from)
                return muse_set_error(MUSE_ERR_HIP, "muse_run_sharded: the workgroups of the loop kernel were not all resident at once on "
                                                    "some rank; later calls run the host loop");
            // ... a cold start is simply run again, by the host loop below: the same bits
        }
    }
    StepParams sp;
    memset(&sp, 0, sizeof sp);
    sp.ntheta = nt;
    sp.nsims = S;
    sp.prior_kind = o->prior_kind;
    sp.alpha = o->alpha;
    sp.theta_rtol = o->theta_rtol;
    for (int k = 0; k < nt; ++k) {
        sp.prior_mean[k] = o->prior_mean[k];
        sp.prior_sigma[k] = o->prior_sigma[k];
    }
    StepWork work;
    double theta[kMaxTheta], theta_next[kMaxTheta], mean[kMaxTheta], var[kMaxTheta];
    for (int k = 0; k < nt; ++k) theta[k] = theta0[k];
    std::vector<double> gall((size_t)world * rows * nt), g((size_t)(S + 1) * nt);
    std::vector<muse_info> info((size_t)nlocal);
    auto now_s = [] { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-9; };
    int n = 0;
    for (int i = 1; i <= o->maxsteps; ++i) {
        const double t_start = now_s();
        if (i > 2) {  // convergence on the last two records (src/muse.jl:163-166); a NaN compares false and the loop goes on
            const int cv = step_converged(nt, hist_out + (int64_t)(i - 2) * H, hist_out + (int64_t)(i - 3) * H, o->theta_rtol);
            if (cv < 0) return muse_set_error(MUSE_ERR_INVALID, "muse_run_sharded: DomainError in the convergence test: dtheta' H^-1_post' dtheta > 0 (H^-1_post' is not negative definite)");
            if (cv > 0) break;
        }
        const int z0_mode = (i > 1 || o->z0_warm) ? MUSE_Z0_WARM : MUSE_Z0_ZERO;
        rc = muse_map_and_score_batch_gather_async(ctx, seed, lo, hi, rank == 0 ? 1 : 0, theta, o->atol, z0_mode, rows, 0);
        if (rc) return rc;
        rc = muse_batch_wait_gathered(ctx, 0, gall.data(), info.data());
        if (rc) return rc;
        // every score in the reference's order: the data element (rank 0's first row), then the simulations by rank
        {
            double* out = g.data();
            for (int r = 0; r < world; ++r) {
                int64_t l, h;
                block(r, l, h);
                const int64_t cnt = (h - l) + (r == 0 ? 1 : 0);
                memcpy(out, gall.data() + (size_t)r * rows * nt, (size_t)cnt * nt * sizeof(double));
                out += cnt * nt;
            }
        }
        double* h = hist_out + (int64_t)(i - 1) * H;
        double* gs = gsims_out + (int64_t)(i - 1) * S * nt;
        memcpy(gs, g.data() + nt, (size_t)S * nt * sizeof(double));
        if (info_out) memcpy(info_out + (int64_t)(i - 1) * nlocal, info.data(), (size_t)nlocal * sizeof(muse_info));
        for (int k = 0; k < nt; ++k) step_moments(k, nt, S, gs, mean[k], var[k]);
        const int err = step_record(sp, theta, g.data(), mean, var, h, theta_next, work);
        if (err == STEP_SINGULAR_LIKE) return muse_set_error(MUSE_ERR_INVALID, "muse_run: singular H^-1_like (zero score variance)");
        if (err == STEP_SINGULAR_POST) return muse_set_error(MUSE_ERR_INVALID, "muse_run: singular posterior Hessian");
        for (int k = 0; k < nt; ++k) theta[k] = theta_next[k];
        h[7 * nt + nt * nt] = now_s() - t_start;
        n = i;
    }
    *niter_out = n;
    for (int k = 0; k < nt; ++k) theta_out[k] = theta[k];
    return MUSE_OK;
}
