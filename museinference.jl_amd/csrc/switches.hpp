// switches.hpp -- every environment switch of the library, in one place.
//
// The environment is read ONCE per context, by muse_ctx_create (Switches::from_environment); no call path after that calls
// getenv.  The communicator of a context (muse_comm.cpp) uses its context's copy.  What a test or a tuning run wants to change
// on a LIVE context goes through muse_debug_flags (include/muse_hip.h, "Diagnostics"): bits 16-20 there are the run-time forms of
// host_board, sharded_host_loop, loop_oversubscribe, run_timing and fd_fold.
//
// None of these changes a result bit (they choose between code paths that tests hold bit-equal) except cluster_size, which
// -- like muse_set_element_split -- changes the summation tree of a cluster placement.
#pragma once
#include <stdlib.h>

namespace muse {

struct Switches {
    // ---- placement / launch geometry (tuning aids)
    bool no_lds_s = false;              // MUSE_DEBUG_NO_LDS_S: stencil model keeps the search direction in HBM
    bool no_xcd_local = false;          // MUSE_DEBUG_NO_XCD_LOCAL: clusters of the elementwise models span the XCDs
    bool no_big_tier = false;           // MUSE_DEBUG_NO_BIG_TIER: 2-8 components in a streaming placement run the small tiers' kernels
    bool no_ext_launch = false;         // MUSE_DEBUG_NO_EXT_LAUNCH: a result area's completion event is recorded behind the launch
    bool fd_fold = false;               // MUSE_FD_FOLD: get_H!'s finite-difference map as ONE launch that carries its fiducial MAP (bit 20; built
                                        // and measured in round 6: 8 us SLOWER per 513-problem call than the two launches -- off by default)
    bool no_fid_normals = false;        // MUSE_DEBUG_NO_FID_NORMALS: get_H!'s fiducial MAP draws its own normals (bit 21; round 6: a kernel of
                                        // its own draws them with the whole GPU -- the fiducial is the one serial problem of the call)
    int cluster_size = 0;               // MUSE_DEBUG_CLUSTER_SIZE=k: workgroups per element of the streaming clusters (0: by N and model)
    int shared_gpu_ranks = 1;           // MUSE_SHARED_GPU_RANKS=n: n processes share this GPU; cluster launches take 1/n of the compute units
    // ---- normals cache
    bool no_ncache = false;             // MUSE_DEBUG_NO_NCACHE
    long long ncache_max_bytes = 8192ll << 20;   // MUSE_NCACHE_MAX_MB
    // ---- the native muse! loops
    bool no_loop_kernel = false;        // MUSE_DEBUG_NO_LOOP_KERNEL: muse_run_device runs the host loop
    bool loop_any_ntheta = false;       // MUSE_DEBUG_LOOP_ANY_NTHETA: the loop kernel whatever ntheta (tests, fuzz_loops.py)
    bool loop_any_scratch = false;      // MUSE_DEBUG_LOOP_ANY_SCRATCH: the loop kernel however much of its state the compiler spilled
    bool loop_dedicated_stepper = false;   // MUSE_DEBUG_LOOP_DEDICATED_STEPPER: the stepper never owns elements (debug flag bit 7 likewise)
    bool loop_oversubscribe = false;    // MUSE_DEBUG_LOOP_OVERSUBSCRIBE: test hook, more workgroups than are resident at once (bit 18)
    int loop_grid = 0;                  // MUSE_DEBUG_LOOP_GRID=n: the loop kernel with n workers (0: as many as are resident)
    bool run_timing = false;            // MUSE_DEBUG_RUN_TIMING: the native loops say on stderr which loop ran and what it cost (bit 19)
    // ---- the exchange between ranks (muse_comm.cpp)
    bool no_board = false;              // MUSE_DEBUG_NO_BOARD: no score board at all (the sharded loop is host-driven)
    bool no_ipc_board = false;          // MUSE_DEBUG_NO_IPC_BOARD: no boards in device memory (hipIpc)
    bool host_board = false;            // MUSE_DEBUG_HOST_BOARD: the sharded loop uses the board in pinned host memory (bit 16)
    bool sharded_host_loop = false;     // MUSE_DEBUG_SHARDED_HOST_LOOP: muse_run_sharded runs the host-driven loop (bit 17)
    bool comm_one_stream = false;       // MUSE_COMM_ONE_STREAM: RCCL collectives in line with the solver
    bool comm_direct_host = false;      // MUSE_COMM_DIRECT_HOST: RCCL receives straight into pinned host memory
    double shm_timeout_s = 0.0;         // MUSE_SHM_TIMEOUT_S: bound of every wait of the shared-memory transport (0: its default, 60 s)
    double handshake_ms = 0.0;          // MUSE_BOARD_HANDSHAKE_MS: bound of the boards' set-up hand-shake (0: its default, 50 ms)
    int handshake_fail = 0;             // MUSE_DEBUG_HANDSHAKE_FAIL=1|2|3: test hook -- the hand-shake of the device boards (1), of the host board
                                        // (2) or of both (3) stores its granules beside the slots it polls: the verdict is "not seen", as it
                                        // would be if stores into a peer's board never became visible to the peer

    static Switches from_environment() {
        Switches s;
        auto on = [](const char* name) { return getenv(name) != nullptr; };
        auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        s.no_lds_s = on("MUSE_DEBUG_NO_LDS_S");
        s.no_xcd_local = on("MUSE_DEBUG_NO_XCD_LOCAL");
        s.no_big_tier = on("MUSE_DEBUG_NO_BIG_TIER");
        s.no_ext_launch = on("MUSE_DEBUG_NO_EXT_LAUNCH");
        s.fd_fold = on("MUSE_FD_FOLD");
        s.no_fid_normals = on("MUSE_DEBUG_NO_FID_NORMALS");
        s.cluster_size = num("MUSE_DEBUG_CLUSTER_SIZE", 0);
        s.shared_gpu_ranks = num("MUSE_SHARED_GPU_RANKS", 1);
        s.no_ncache = on("MUSE_DEBUG_NO_NCACHE");
        if (const char* e = getenv("MUSE_NCACHE_MAX_MB")) s.ncache_max_bytes = atoll(e) << 20;
        s.no_loop_kernel = on("MUSE_DEBUG_NO_LOOP_KERNEL");
        s.loop_any_ntheta = on("MUSE_DEBUG_LOOP_ANY_NTHETA");
        s.loop_any_scratch = on("MUSE_DEBUG_LOOP_ANY_SCRATCH");
        s.loop_dedicated_stepper = on("MUSE_DEBUG_LOOP_DEDICATED_STEPPER");
        s.loop_oversubscribe = on("MUSE_DEBUG_LOOP_OVERSUBSCRIBE");
        s.loop_grid = num("MUSE_DEBUG_LOOP_GRID", 0);
        s.run_timing = on("MUSE_DEBUG_RUN_TIMING");
        s.no_board = on("MUSE_DEBUG_NO_BOARD");
        s.no_ipc_board = on("MUSE_DEBUG_NO_IPC_BOARD");
        s.host_board = on("MUSE_DEBUG_HOST_BOARD");
        s.sharded_host_loop = on("MUSE_DEBUG_SHARDED_HOST_LOOP");
        s.comm_one_stream = on("MUSE_COMM_ONE_STREAM");
        s.comm_direct_host = on("MUSE_COMM_DIRECT_HOST");
        if (const char* e = getenv("MUSE_SHM_TIMEOUT_S")) s.shm_timeout_s = atof(e) > 0 ? atof(e) : 0.0;
        if (const char* e = getenv("MUSE_BOARD_HANDSHAKE_MS")) s.handshake_ms = atof(e) > 0 ? atof(e) : 0.0;
        s.handshake_fail = num("MUSE_DEBUG_HANDSHAKE_FAIL", 0);
        return s;
    }
};

// Host-side bits of muse_debug_flags (bits 0-9 travel to the kernels in BatchArgs::debug).
enum : int {
    kDebugHostBoard = 1 << 16,          // the sharded loop's scores meet on the board in pinned host memory
    kDebugShardedHostLoop = 1 << 17,    // muse_run_sharded runs the host-driven loop
    kDebugLoopOversubscribe = 1 << 18, // test hook: a loop launch with more workgroups than can be resident at once
    kDebugRunTiming = 1 << 19,         // the native loops report on stderr which loop ran
    kDebugFdFold = 1 << 20,            // get_H!'s finite-difference map as one launch that carries its fiducial MAP
    kDebugNoFidNormals = 1 << 21,      // get_H!'s fiducial MAP draws its own normals
};

}  // namespace muse
