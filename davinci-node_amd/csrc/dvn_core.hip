// libdvn_prover.so -- context lifetime, memory helpers, error plumbing (include/dvn_prover.h).
// Host side of the boundary described in SURVEY.md §8(b); stands where gnark's icicle backend stands behind
// /root/reference/prover/prover_gpu.go:24-61.
#include "dvn_internal.h"

thread_local std::string tl_error;
int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    tl_error = buf;
    return code;
}

// ------------------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------------------
extern "C" const char* dvn_version(void) { return "dvn-prover 0.1 (gfx950, BN254, 9x29-bit Montgomery)"; }
extern "C" const char* dvn_last_error(const dvn_ctx*) { return tl_error.c_str(); }

extern "C" int32_t dvn_init(const int* devs, int ndev, dvn_ctx** out) {
    if (!out) return fail(DVN_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) { (void)hipGetLastError(); return fail(DVN_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e)); }
    int dev = (devs && ndev > 0) ? devs[0] : 0;
    if (dev < 0 || dev >= count) return fail(DVN_ERR_NO_DEVICE, "device ordinal %d out of range (%d devices)", dev, count);
    dvn_ctx* ctx = new (std::nothrow) dvn_ctx();
    if (!ctx) return fail(DVN_ERR_OOM, "host allocation failed");
    ctx->device = dev;
    const char* ff = getenv("DVN_FORCE_FAIL");
    ctx->force_fail = ff && (ff[0] == '1' || ff[0] == 'y' || ff[0] == 't');
    hipError_t e2 = hipSetDevice(dev);
    if (e2 == hipSuccess) e2 = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e2 == hipSuccess; ++i) e2 = hipEventCreate(&ctx->ev[i]);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&ctx->ev_dep, hipEventDisableTiming);
    for (int l = 0; l < DVN_LANES && e2 == hipSuccess; ++l) {
        MsmLane& L = ctx->lane[l];
        e2 = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
        for (int i = 0; i < 4 && e2 == hipSuccess; ++i) e2 = hipEventCreate(&L.ev[i]);
        if (e2 == hipSuccess) e2 = hipHostMalloc((void**)&L.h_sums, DVN_HSUMS_WORDS * 4, hipHostMallocDefault);
    }
    if (e2 != hipSuccess) { delete ctx; (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "context setup failed: %s", hipGetErrorString(e2)); }
    { int32_t rc = dvn_internal_static_scratch_init(ctx); if (rc != DVN_OK) { dvn_shutdown(ctx); return rc; } }
    *out = ctx;
    return DVN_OK;
}
extern "C" void dvn_shutdown(dvn_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (int l = 0; l < DVN_LANES; ++l) {
        MsmLane& L = ctx->lane[l];
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        if (L.ws) (void)hipFree(L.ws);
        if (L.h_sums) (void)hipHostFree(L.h_sums);
        for (int i = 0; i < 4; ++i) if (L.ev[i]) (void)hipEventDestroy(L.ev[i]);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (ctx->ev_dep) (void)hipEventDestroy(ctx->ev_dep);
    for (int i = 0; i < 4; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
extern "C" int32_t dvn_malloc(dvn_ctx* ctx, size_t bytes, void** dptr) {
    TRY(enter(ctx));
    if (!dptr) return fail(DVN_ERR_INVALID_ARG, "dptr is NULL");
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return DVN_OK;
}
extern "C" int32_t dvn_free(dvn_ctx* ctx, void* dptr) { TRY(enter(ctx)); if (dptr) HIP_TRY(hipFree(dptr)); return DVN_OK; }
extern "C" int32_t dvn_memcpy_h2d(dvn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    TRY(enter(ctx)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return DVN_OK;
}
extern "C" int32_t dvn_memcpy_d2h(dvn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    TRY(enter(ctx)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return DVN_OK;
}
extern "C" int32_t dvn_synchronize(dvn_ctx* ctx) { TRY(enter(ctx)); HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipDeviceSynchronize()); return DVN_OK; }
extern "C" int32_t dvn_last_kernel_ms(dvn_ctx* ctx, int which, float* ms) {
    if (!ctx || !ms || which < 0 || which > 1) return fail(DVN_ERR_INVALID_ARG, "bad argument");
    *ms = ctx->last_ms[which];
    return DVN_OK;
}

