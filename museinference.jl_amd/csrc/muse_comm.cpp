// muse_comm.cpp -- RCCL exchange of the per-sim accumulators (collectives C1-C3 of SURVEY.md §2).
//
// The reference gathers map results to the master process through Distributed.pmap
// (src/util.jl:74-83) and reduces them there (src/muse.jl:183,188,446,529).  Here every rank owns a
// contiguous block of sims on its own GPU and the per-rank score blocks / H accumulators are
// exchanged with ONE small RCCL collective per outer iteration over xGMI.  Messages are <= 64 KB,
// so the cost is collective latency, not link bandwidth.
//
// librccl is opened lazily (dlopen) so that libmuse_hip.so loads on hosts without a usable RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>

#include "../../include/muse_hip.h"

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
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
thread_local std::string g_comm_err;

bool load_rccl() {
    if (g_rccl.h) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.AllReduce)
        return false;
    g_rccl.h = h;
    return true;
}
}  // namespace

// accessors implemented in muse_engine.hip (the context layout is private to that file)
extern "C" {
int muse_ctx_comm_slot(muse_ctx* ctx, void*** comm, int* device, void** stream);
int muse_ctx_comm_buffer(muse_ctx* ctx, size_t doubles, double** buf);
int muse_set_error(int code, const char* msg);
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

extern "C" {

int muse_comm_unique_id(void* id_out) {
    if (!id_out) return muse_set_error(MUSE_ERR_INVALID, "id_out is NULL");
    if (!load_rccl()) return muse_set_error(MUSE_ERR_RCCL, "librccl could not be loaded");
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, MUSE_UNIQUE_ID_BYTES);
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
    if (!load_rccl()) return muse_set_error(MUSE_ERR_RCCL, "librccl could not be loaded");
    HIPCHK2(hipSetDevice(device));
    ncclUniqueId uid;
    memcpy(&uid, id, MUSE_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&comm, nranks, uid, rank));
    *slot = comm;
    return MUSE_OK;
}

int muse_comm_destroy(muse_ctx* ctx) {
    void** slot;
    int device;
    void* stream;
    int rc = muse_ctx_comm_slot(ctx, &slot, &device, &stream);
    if (rc) return rc;
    if (*slot && g_rccl.h) g_rccl.CommDestroy((ncclComm_t)*slot);
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
    typedef ncclResult_t (*CountFn)(ncclComm_t, int*);
    CountFn cnt = (CountFn)dlsym(g_rccl.h, "ncclCommCount");
    if (!cnt) return muse_set_error(MUSE_ERR_RCCL, "ncclCommCount missing");
    RCCLCHK(cnt((ncclComm_t)*slot, &nranks));
    double* buf;
    rc = muse_ctx_comm_buffer(ctx, (size_t)count * (size_t)(nranks + 1), &buf);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK2(hipMemcpyAsync(buf, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
    RCCLCHK(g_rccl.AllGather(buf, buf + count, (size_t)count, ncclFloat64, (ncclComm_t)*slot, st));
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
    double* buf;
    rc = muse_ctx_comm_buffer(ctx, (size_t)count, &buf);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK2(hipMemcpyAsync(buf, hostbuf, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
    RCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)*slot, st));
    HIPCHK2(hipMemcpyAsync(hostbuf, buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK2(hipStreamSynchronize(st));
    return MUSE_OK;
}

}  // extern "C"
