// Per-curve entry points shared by the pairing-curve instantiations other than BN254 (BLS12-377, BLS12-381): thin
// templates over (G1, G2, scalar config); each curve's translation unit supplies its generator and scalar modulus.
#pragma once
#include "msm_driver.h"
#include "ntt_driver.h"

// i-th canonical scalar of SplitMix64 stream `seed`: the 256-bit value of four stream words reduced mod r
// (oracle/curves_py.py Suite.fr_at)
static inline void suite_fr_at_words(const u32 rm[8], u64 seed, u64 i, u32 out[8]) {
    for (int k = 0; k < 4; ++k) { u64 v = splitmix_at(seed, 4 * i + k); out[2 * k] = (u32)v; out[2 * k + 1] = (u32)(v >> 32); }
    while (geq8(out, rm)) sub8(out, rm);
}
// P_i = [s0]G + i*[s1]G, (s0, s1) = first two scalars of stream `seed`: bench/test input generator
template <class Cv> static int32_t suite_points_synthetic(dvn_ctx* ctx, int curve, const typename Cv::A& gen, const u32 rm[8], uint64_t seed, size_t n, dvn_points** out) {
    TRY(enter(ctx));
    if (!out) return fail(DVN_ERR_INVALID_ARG, "out is NULL");
    if (n >= (1ull << 31)) return fail(DVN_ERR_INVALID_ARG, "point set too large");
    std::lock_guard<std::mutex> lk(ctx->mu);
    u32 s0[8], s1[8]; suite_fr_at_words(rm, seed, 0, s0); suite_fr_at_words(rm, seed, 1, s1);
    typename Cv::A base = Cv::to_affine(host_scalar_mul<Cv>(gen, s0)), step = Cv::to_affine(host_scalar_mul<Cv>(gen, s1));
    return points_progression<Cv>(ctx, n, curve, base, step, out);
}
template <class G1T, class G2T> static int32_t suite_points_download(dvn_ctx* ctx, const char* name, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) {
    TRY(enter(ctx));
    if (!pts || (!xy && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->suite != SuiteOf<G1T>::value) return fail(DVN_ERR_INVALID_ARG, "point set is not %s", name);
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return pts->curve == 1 ? points_download<G1T>(ctx, pts, off, n, xy) : points_download<G2T>(ctx, pts, off, n, xy);
}
template <class S> static int32_t suite_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) {
    TRY(enter(ctx));
    if (!d_out && n) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (kind != 0 && kind != 1) return fail(DVN_ERR_INVALID_ARG, "kind must be 0 (uniform) or 1 (witness-like)");
    if (n == 0) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    k_scalars_synthetic<S><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((u32*)d_out, n, seed, kind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DVN_OK;
}
template <class S> static int32_t suite_ntt(dvn_ctx* ctx, const char* name, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) {
    TRY(enter(ctx));
    if (!dom || !data) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (dom->suite != DomainConsts<S>::SUITE) return fail(DVN_ERR_INVALID_ARG, "domain is not a %s domain", name);
    if (decimation != DVN_DIF && decimation != DVN_DIT) return fail(DVN_ERR_INVALID_ARG, "decimation must be DVN_DIF or DVN_DIT");
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t bytes = ((size_t)1 << dom->log_n) * S::EW * 4;
    u32* d_data;
    const bool on_dev = flags & DVN_NTT_DATA_ON_DEVICE;
    if (on_dev) d_data = (u32*)data;
    else {
        TRY(ensure_ws(ctx, bytes + 256));
        d_data = (u32*)ctx->ws;
        HIP_TRY(hipMemcpyAsync(d_data, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    TRY(ntt_device<S>(ctx, dom, d_data, inverse != 0, decimation, on_coset != 0));
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (!on_dev) HIP_TRY(hipMemcpyAsync(data, d_data, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipEventElapsedTime(&ctx->last_ms[0], ctx->ev[0], ctx->ev[1]));
    ctx->last_ms[1] = ctx->last_ms[0];
    return DVN_OK;
}
