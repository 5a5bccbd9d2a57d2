// fft.Domain, FFT / FFTInverse and groth16 computeH entry points (BN254 scalar field).
#include "ntt_driver.h"

extern "C" int32_t dvn_ntt_domain_free(dvn_ctx* ctx, dvn_domain* d) try {
    TRY(enter(ctx));
    if (!d) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipStreamSynchronize(ctx->stream);
    ntt_domain_release(d);
    return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) try { return ntt_domain_create<FrCfg>(ctx, log_n, out); } DVN_CATCH

extern "C" int32_t dvn_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) try {
    TRY(enter(ctx));
    if (!dom || !data) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (dom->suite != 0) return fail(DVN_ERR_INVALID_ARG, "domain belongs to another curve");
    if (decimation != DVN_DIF && decimation != DVN_DIT) return fail(DVN_ERR_INVALID_ARG, "decimation must be DVN_DIF or DVN_DIT");
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t bytes = ((size_t)1 << dom->log_n) * 32;
    u32* d_data;
    const bool on_dev = flags & DVN_NTT_DATA_ON_DEVICE;
    if (on_dev) d_data = (u32*)data;
    else {
        TRY(ensure_ws(ctx, bytes + 256));
        d_data = (u32*)ctx->ws;
        HIP_TRY(hipMemcpyAsync(d_data, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    TRY(ntt_device<FrCfg>(ctx, dom, d_data, inverse != 0, decimation, on_coset != 0));
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (!on_dev) HIP_TRY(hipMemcpyAsync(data, d_data, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipEventElapsedTime(&ctx->last_ms[0], ctx->ev[0], ctx->ev[1]));
    ctx->last_ms[1] = ctx->last_ms[0];
    return DVN_OK;
} DVN_CATCH

int32_t dvn_internal_compute_h_dev(dvn_ctx* ctx, const dvn_domain* dom, u32* da, u32* db, u32* dc) { return compute_h_device<FrCfg>(ctx, dom, da, db, dc); }
uint32_t dvn_internal_domain_log_n(const dvn_domain* dom) { return dom->log_n; }

extern "C" int32_t dvn_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                         size_t len, uint64_t* h_out, uint32_t flags) try {
    return compute_h_api<FrCfg>(ctx, "BN254", dom, a, b, c, len, h_out, flags);
} DVN_CATCH
