// fft.Domain, FFT / FFTInverse and groth16 computeH entry points (BN254 scalar field).
#include "ntt_driver.h"

extern "C" int32_t dvn_ntt_domain_free(dvn_ctx* ctx, dvn_domain* d) {
    TRY(enter(ctx));
    if (!d) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipStreamSynchronize(ctx->stream);
    u32* ptrs[] = {d->tw_fwd, d->tw_inv, d->g_lo, d->g_hi, d->gi_lo, d->gi_hi, d->card_inv, d->h_consts};
    for (u32* p : ptrs) if (p) (void)hipFree(p);
    delete d;
    return DVN_OK;
}
extern "C" int32_t dvn_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) { return ntt_domain_create<FrCfg>(ctx, log_n, out); }

extern "C" int32_t dvn_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) {
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
}

// computeH core on device buffers (gnark prove.go computeH): 3 x FFTInverse(DIF), 3 x FFT(DIT, coset),
// pointwise (a*b - c) * den, FFTInverse(DIF, coset).  Result in da, bit-reversed.
int32_t dvn_internal_compute_h_dev(dvn_ctx* ctx, const dvn_domain* dom, u32* da, u32* db, u32* dc) {
    const size_t n = (size_t)1 << dom->log_n;
    u32* v[3] = {da, db, dc};
    for (int i = 0; i < 3; ++i) TRY(ntt_device<FrCfg>(ctx, dom, v[i], 1, DVN_DIF, 0));
    for (int i = 0; i < 3; ++i) TRY(ntt_device<FrCfg>(ctx, dom, v[i], 0, DVN_DIT, 1));
    k_h_pointwise<FrCfg><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(da, db, dc, dom->h_consts, n);
    HIP_TRY(hipGetLastError());
    TRY(ntt_device<FrCfg>(ctx, dom, da, 1, DVN_DIF, 1));
    return DVN_OK;
}
uint32_t dvn_internal_domain_log_n(const dvn_domain* dom) { return dom->log_n; }

extern "C" int32_t dvn_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                         size_t len, uint64_t* h_out, uint32_t flags) {
    TRY(enter(ctx));
    if (!dom || !a || !b || !c || !h_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const size_t n = (size_t)1 << dom->log_n;
    if (len > n) return fail(DVN_ERR_INVALID_ARG, "len = %zu exceeds domain size %zu", len, n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const bool on_dev = flags & DVN_NTT_DATA_ON_DEVICE;
    const hipMemcpyKind in_kind = on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    TRY(ensure_ws(ctx, 3 * n * 32 + 1024));
    u32* da = (u32*)ctx->ws; u32* db = da + n * 8; u32* dc = db + n * 8;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemsetAsync(da, 0, 3 * n * 32, st));
    HIP_TRY(hipMemcpyAsync(da, a, len * 32, in_kind, st));
    HIP_TRY(hipMemcpyAsync(db, b, len * 32, in_kind, st));
    HIP_TRY(hipMemcpyAsync(dc, c, len * 32, in_kind, st));
    HIP_TRY(hipEventRecord(ctx->ev[0], st));
    TRY(dvn_internal_compute_h_dev(ctx, dom, da, db, dc));
    HIP_TRY(hipEventRecord(ctx->ev[1], st));
    HIP_TRY(hipMemcpyAsync(h_out, da, n * 32, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&ctx->last_ms[0], ctx->ev[0], ctx->ev[1]));
    ctx->last_ms[1] = ctx->last_ms[0];
    return DVN_OK;
}
