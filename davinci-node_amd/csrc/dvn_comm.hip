// libdvn_prover.so -- the exchange step of the sharded provers for hosts that run ONE PROCESS PER GPU without torch:
// an RCCL communicator behind four C entry points (include/dvn_prover.h, dvn_comm_*).  The ranks all-gather their partial
// results (96 bytes per MSM share, 576 per proof share) over xGMI and add them up locally (dvn_msm_g1_sum,
// dvn_groth16_prove_combine): RCCL has no elliptic-curve reduction operator, so "all-reduce" is all-gather + a local sum.
// The reference has no counterpart (its ICICLE backend drives one GPU: /root/reference/prover/prover_gpu.go:24-61); the
// serialisation point this shortens is /root/reference/sequencer/statetransition.go:80.
// librccl is opened at the first use, not linked: a single-GPU host never loads it.
#include <dlfcn.h>
#include "dvn_internal.h"

namespace {
constexpr int RCCL_UNIQUE_ID_BYTES = 128;                  // rccl.h NCCL_UNIQUE_ID_BYTES
constexpr int RCCL_UINT64 = 5;                             // rccl.h ncclUint64
struct RcclId { char internal[RCCL_UNIQUE_ID_BYTES]; };
typedef void* RcclComm;
struct Rccl {
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    Rccl() {
        for (const char* lib : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
            CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
            CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
            AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
            GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
            ok = GetUniqueId && CommInitRank && CommDestroy && AllGather && GetErrorString;
            if (ok) return;
        }
    }
};
Rccl& rccl() { static Rccl r; return r; }
int32_t rccl_fail(const char* what, int rc) {
    return fail(DVN_ERR_DEVICE, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
}
}  // namespace

struct dvn_comm {
    RcclComm comm = nullptr;
    uint32_t rank = 0, world = 0;
    int device = 0;
    u32* d_send = nullptr; u32* d_recv = nullptr;          // staging: DVN_COMM_MAX_WORDS and world x DVN_COMM_MAX_WORDS uint64
};

extern "C" int32_t dvn_comm_unique_id(uint8_t id[128]) try {
    if (!id) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (!rccl().ok) return fail(DVN_ERR_DEVICE, "librccl could not be loaded");
    RcclId u;
    const int rc = rccl().GetUniqueId(&u);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, RCCL_UNIQUE_ID_BYTES);
    return DVN_OK;
} DVN_CATCH

extern "C" int32_t dvn_comm_init(dvn_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t world, dvn_comm** out) try {
    TRY(enter(ctx));
    if (!id || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (world == 0 || rank >= world || world > 1024) return fail(DVN_ERR_INVALID_ARG, "rank %u outside world %u", rank, world);
    if (!rccl().ok) return fail(DVN_ERR_DEVICE, "librccl could not be loaded");
    std::lock_guard<std::mutex> lk(ctx->mu);
    dvn_comm* c = new dvn_comm();
    c->rank = rank; c->world = world; c->device = ctx->device;
    RcclId u; memcpy(u.internal, id, RCCL_UNIQUE_ID_BYTES);
    int rc = rccl().CommInitRank(&c->comm, (int)world, u, (int)rank);          // collective: returns when every rank has called it
    if (rc != 0) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    hipError_t e = hipMalloc((void**)&c->d_send, (size_t)DVN_COMM_MAX_WORDS * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_recv, (size_t)world * DVN_COMM_MAX_WORDS * 8);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (c->d_send) (void)hipFree(c->d_send);
        (void)rccl().CommDestroy(c->comm); delete c;
        return fail(DVN_ERR_OOM, "staging buffers of the communicator: %s", hipGetErrorString(e));
    }
    *out = c;
    return DVN_OK;
} DVN_CATCH

// every rank contributes `words` uint64 (host memory); every rank receives world x words, rank r's at [r * words, (r + 1) * words)
extern "C" int32_t dvn_comm_all_gather(dvn_ctx* ctx, dvn_comm* c, const uint64_t* local, size_t words, uint64_t* all) try {
    TRY(enter(ctx));
    if (!c || !local || !all) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (c->device != ctx->device) return fail(DVN_ERR_INVALID_ARG, "communicator belongs to device %d, context to device %d", c->device, ctx->device);
    if (words == 0 || words > DVN_COMM_MAX_WORDS) return fail(DVN_ERR_INVALID_ARG, "all-gather of %zu words: between 1 and %d per rank", words, DVN_COMM_MAX_WORDS);
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(c->d_send, local, words * 8, hipMemcpyHostToDevice, st));
    const int rc = rccl().AllGather(c->d_send, c->d_recv, words, RCCL_UINT64, c->comm, st);
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    HIP_TRY(hipMemcpyAsync(all, c->d_recv, (size_t)c->world * words * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DVN_OK;
} DVN_CATCH

extern "C" int32_t dvn_comm_free(dvn_ctx* ctx, dvn_comm* c) try {
    if (!c) return DVN_OK;
    TRY(enter(ctx));
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipStreamSynchronize(ctx->stream);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    const int rc = c->comm ? rccl().CommDestroy(c->comm) : 0;
    delete c;
    if (rc != 0) return rccl_fail("ncclCommDestroy", rc);
    return DVN_OK;
} DVN_CATCH
