// fft.Domain, FFT / FFTInverse and groth16 computeH entry points.
#include "dvn_internal.h"
#include "ntt_kernels.h"

// ------------------------------------------------------------------------------------------------------
// NTT
// ------------------------------------------------------------------------------------------------------
static Fr fr_from_u64(u64 v) {
    u32 w[8] = {(u32)v, (u32)(v >> 32), 0, 0, 0, 0, 0, 0};
    return fe_from_canonical<FrCfg>(fe_unpack<FrCfg>(w));
}
static Fr fr_root_of_unity(u32 log_n) {
    // w_{2^28} = 5^((r-1)/2^28)  (SURVEY.md row a13); square down to order 2^log_n
    static const u64 root[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL};
    Fr w = fe_from_canonical<FrCfg>(fe_unpack<FrCfg>(*(const u32(*)[8])root));
    for (u32 i = log_n; i < 28; ++i) w = fe_sqr<FrCfg>(w);
    return w;
}
static void fr_store_internal(const Fr& v, u32* w) { u32 t[8]; fe_pack<FrCfg>(fe_canonical<FrCfg>(v), t); memcpy(w, t, 32); }

static int32_t launch_powers(dvn_ctx* ctx, u32* d_out, const Fr& base, const Fr& first, size_t count, u32* d_scratch16) {
    u32 h[16]; fr_store_internal(base, h); fr_store_internal(first, h + 8);
    HIP_TRY(hipMemcpyAsync(d_scratch16, h, 64, hipMemcpyHostToDevice, ctx->stream));
    const u32 chunk = 64;
    size_t threads = (count + chunk - 1) / chunk;
    k_powers<<<(unsigned)((threads + 63) / 64), 64, 0, ctx->stream>>>(d_out, d_scratch16, count, chunk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // h goes out of scope
    return DVN_OK;
}

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
extern "C" int32_t dvn_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) {
    TRY(enter(ctx));
    if (!out) return fail(DVN_ERR_INVALID_ARG, "out is NULL");
    if (log_n < 1 || log_n > 27) return fail(DVN_ERR_INVALID_ARG, "log_n = %u outside [1, 27]", log_n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    dvn_domain* d = new (std::nothrow) dvn_domain();
    if (!d) return fail(DVN_ERR_OOM, "host allocation failed");
    d->log_n = log_n;
    const size_t n = (size_t)1 << log_n;
    d->lo_bits = log_n < 10 ? log_n : 10;
    const size_t nlo = (size_t)1 << d->lo_bits, nhi = n >> d->lo_bits;
    u32* scratch = nullptr;
    int32_t rc = DVN_OK;
    auto alloc = [&](u32** p, size_t entries) -> int32_t { HIP_TRY(hipMalloc((void**)p, entries * 32)); return DVN_OK; };
    do {
        if ((rc = alloc(&d->tw_fwd, n)) || (rc = alloc(&d->tw_inv, n)) || (rc = alloc(&d->g_lo, nlo)) || (rc = alloc(&d->g_hi, nhi)) ||
            (rc = alloc(&d->gi_lo, nlo)) || (rc = alloc(&d->gi_hi, nhi)) || (rc = alloc(&d->card_inv, 1)) || (rc = alloc(&d->h_consts, 2)) ||
            (rc = alloc(&scratch, 2))) break;
        const Fr one = fe_one<FrCfg>();
        const Fr w = fr_root_of_unity(log_n), wi = fe_inv<FrCfg>(w);
        Fr ws = w, wis = wi;
        for (u32 s = 0; s < log_n && rc == DVN_OK; ++s) {
            const size_t half = n >> (s + 1), offs = n - (n >> s);
            if ((rc = launch_powers(ctx, d->tw_fwd + offs * 8, ws, one, half, scratch))) break;
            if ((rc = launch_powers(ctx, d->tw_inv + offs * 8, wis, one, half, scratch))) break;
            ws = fe_sqr<FrCfg>(ws); wis = fe_sqr<FrCfg>(wis);
        }
        if (rc) break;
        const Fr g = fr_from_u64(5), gi = fe_inv<FrCfg>(g), ninv = fe_inv<FrCfg>(fr_from_u64(n));
        if ((rc = launch_powers(ctx, d->g_lo, g, one, nlo, scratch))) break;
        if ((rc = launch_powers(ctx, d->g_hi, fe_pow_u64<FrCfg>(g, nlo), one, nhi, scratch))) break;
        if ((rc = launch_powers(ctx, d->gi_lo, gi, ninv, nlo, scratch))) break;
        if ((rc = launch_powers(ctx, d->gi_hi, fe_pow_u64<FrCfg>(gi, nlo), one, nhi, scratch))) break;
        u32 h[24];
        fr_store_internal(ninv, h);
        // den = 1/(g^n - 1); k1 = den * 2^5 (internal), k2 = den
        const Fr den = fe_inv<FrCfg>(fe_sub<FrCfg>(fe_pow_u64<FrCfg>(g, n), one));
        fr_store_internal(fe_mul<FrCfg>(den, fr_from_u64(32)), h + 8);
        fr_store_internal(den, h + 16);
        hipError_t e = hipMemcpy(d->card_inv, h, 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d->h_consts, h + 8, 64, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = fail(DVN_ERR_DEVICE, "domain constant upload failed: %s", hipGetErrorString(e)); }
    } while (0);
    if (scratch) (void)hipFree(scratch);
    if (rc != DVN_OK) {
        std::string keep = tl_error;
        u32* ptrs[] = {d->tw_fwd, d->tw_inv, d->g_lo, d->g_hi, d->gi_lo, d->gi_hi, d->card_inv, d->h_consts};
        for (u32* p : ptrs) if (p) (void)hipFree(p);
        delete d;
        tl_error = keep;
        return rc;
    }
    *out = d;
    return DVN_OK;
}

// in-place transform of device data; asynchronous on ctx->stream
static int32_t ntt_device(dvn_ctx* ctx, const dvn_domain* d, u32* data, int inverse, int decimation, int coset) {
    const u32 log_n = d->log_n;
    const u32 npass = (log_n + 7) / 8;
    const u32 base = log_n / npass, rem = log_n % npass;
    u32 ks[8];
    for (u32 p = 0; p < npass; ++p) ks[p] = base + (p < rem ? 1 : 0);
    const u32* tw = inverse ? d->tw_inv : d->tw_fwd;
    NttScale none{nullptr, nullptr, 0, 0, 0, 0};
    NttScale sc = none;
    if (!inverse && coset) sc = NttScale{d->g_lo, d->g_hi, d->lo_bits, 2, decimation == DVN_DIT, 1};
    if (inverse && !coset) sc = NttScale{d->card_inv, nullptr, 0, 1, 0, 2};
    if (inverse && coset) sc = NttScale{d->gi_lo, d->gi_hi, d->lo_bits, 2, decimation == DVN_DIF, 2};
    // stage ranges in execution order
    for (u32 p = 0; p < npass; ++p) {
        u32 sa, sb;
        if (decimation == DVN_DIF) { sa = 0; for (u32 q = 0; q < p; ++q) sa += ks[q]; sb = sa + ks[p]; }
        else { sb = log_n; for (u32 q = 0; q < p; ++q) sb -= ks[q]; sa = sb - ks[p]; }
        const u32 K = sb - sa;
        u32 log_i = 3;
        while (K + log_i > log_n) --log_i;                 // tile must not exceed the array
        while (K + log_i > 11) --log_i;                    // <= 2048 elements = 64 KiB of LDS
        const u32 tiles = (u32)(((size_t)1 << log_n) >> (K + log_i));
        const size_t lds = ((size_t)1 << (K + log_i)) * 32;
        NttScale use = none;
        if (sc.mode && ((sc.when == 1 && p == 0) || (sc.when == 2 && p == npass - 1))) use = sc;
        k_ntt_pass<<<tiles, 512, lds, ctx->stream>>>(data, tw, log_n, sa, sb, decimation == DVN_DIT, log_i, use);
        HIP_TRY(hipGetLastError());
    }
    return DVN_OK;
}
extern "C" int32_t dvn_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) {
    TRY(enter(ctx));
    if (!dom || !data) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
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
    TRY(ntt_device(ctx, dom, d_data, inverse != 0, decimation, on_coset != 0));
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
    for (int i = 0; i < 3; ++i) TRY(ntt_device(ctx, dom, v[i], 1, DVN_DIF, 0));
    for (int i = 0; i < 3; ++i) TRY(ntt_device(ctx, dom, v[i], 0, DVN_DIT, 1));
    k_h_pointwise<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(da, db, dc, dom->h_consts, n);
    HIP_TRY(hipGetLastError());
    TRY(ntt_device(ctx, dom, da, 1, DVN_DIF, 1));
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
