// BLS12-377 instantiation: the curve of the voteverifier proofs (60 per batch,
// /root/reference/spec/params/params.go:21-27; GPU dispatch at /root/reference/prover/prover_gpu.go:36-43).
// The same kernels as BN254, instantiated on 14-limb base-field elements (377 bits, external image 6 x u64,
// R = 2^384), Fe2 with u^2 = -5 for G2, and the 253-bit scalar field (two-adicity 47, coset shift 22) for the NTT.
#include "msm_driver.h"
#include "ntt_driver.h"

extern "C" int32_t dvn_bls12377_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) { return points_upload<G1_377>(ctx, xy, n, 1, out); }
extern "C" int32_t dvn_bls12377_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) { return points_upload<G2_377>(ctx, xy, n, 2, out); }
static G1_377::A g1_377_generator() {      // gnark-crypto bls12-377 G1 generator, Montgomery image (oracle/curves_py.py BLS12_377.g1_gen)
    static const u64 g[12] = {0x260f33b9772451f4ULL, 0xc54dd773169d5658ULL, 0x5c1551c469a510ddULL, 0x761662e4425e1698ULL, 0xc97d78cc6f065272ULL, 0x00a41206b361fd4dULL,
                              0x8193961fb8cb81f3ULL, 0x00638d4c5f44adb8ULL, 0xfafaf3dad4daf54aULL, 0xc27849e2d655cd18ULL, 0x2ec3ddb401d52814ULL, 0x007da93326303c71ULL};
    return G1_377::affine_from_ext((const u32*)g);
}
// i-th canonical scalar of SplitMix64 stream `seed` reduced mod r377 (oracle/curves_py.py Suite.fr_at)
static void fr377_at_words(u64 seed, u64 i, u32 out[8]) {
    static const u32 rm[8] = {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};
    for (int k = 0; k < 4; ++k) { u64 v = splitmix_at(seed, 4 * i + k); out[2 * k] = (u32)v; out[2 * k + 1] = (u32)(v >> 32); }
    while (geq8(out, rm)) sub8(out, rm);
}
// P_i = [s0]G + i*[s1]G, (s0, s1) = first two scalars of stream `seed`: bench/test input generator
extern "C" int32_t dvn_bls12377_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out) {
    TRY(enter(ctx));
    if (!out) return fail(DVN_ERR_INVALID_ARG, "out is NULL");
    if (n >= (1ull << 31)) return fail(DVN_ERR_INVALID_ARG, "point set too large");
    std::lock_guard<std::mutex> lk(ctx->mu);
    u32 s0[8], s1[8]; fr377_at_words(seed, 0, s0); fr377_at_words(seed, 1, s1);
    const G1_377::A gen = g1_377_generator();
    G1_377::A base = G1_377::to_affine(host_scalar_mul<G1_377>(gen, s0)), step = G1_377::to_affine(host_scalar_mul<G1_377>(gen, s1));
    u32 hb[2 * G1_377::AW];
    G1_377::store_affine_internal(base, hb); G1_377::store_affine_internal(step, hb + G1_377::AW);
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->suite = 1; p->curve = 1; p->n = n;
    u32* dbs = nullptr;
    hipError_t e = hipMalloc((void**)&p->d, n ? n * G1_377::AW * 4 : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&dbs, sizeof(hb));
    if (e == hipSuccess) e = hipMemcpyAsync(dbs, hb, sizeof(hb), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) {
        const u32 chunk = 16;
        size_t threads = (n + chunk - 1) / chunk;
        k_points_progression<G1_377><<<(unsigned)((threads + 63) / 64), 64, 0, ctx->stream>>>(p->d, n, dbs, chunk);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (dbs) (void)hipFree(dbs);
    if (e != hipSuccess) { if (p->d) (void)hipFree(p->d); delete p; (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "synthetic points failed: %s", hipGetErrorString(e)); }
    *out = p;
    return DVN_OK;
}
extern "C" int32_t dvn_bls12377_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) {
    TRY(enter(ctx));
    if (!pts || (!xy && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->suite != 1) return fail(DVN_ERR_INVALID_ARG, "point set is not BLS12-377");
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return pts->curve == 1 ? points_download<G1_377>(ctx, pts, off, n, xy) : points_download<G2_377>(ctx, pts, off, n, xy);
}
extern "C" int32_t dvn_bls12377_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[18]) {
    return msm_full<G1_377>(ctx, pts, 1, off, n, scalars, flags, out_jac);
}
extern "C" int32_t dvn_bls12377_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[36]) {
    return msm_full<G2_377>(ctx, pts, 2, off, n, scalars, flags, out_jac);
}
extern "C" int32_t dvn_bls12377_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) {
    TRY(enter(ctx));
    if (!d_out && n) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (kind != 0 && kind != 1) return fail(DVN_ERR_INVALID_ARG, "kind must be 0 (uniform) or 1 (witness-like)");
    if (n == 0) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    k_scalars_synthetic<Fr377Cfg><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((u32*)d_out, n, seed, kind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DVN_OK;
}
extern "C" int32_t dvn_bls12377_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) { return ntt_domain_create<Fr377Cfg>(ctx, log_n, out); }
extern "C" int32_t dvn_bls12377_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) {
    TRY(enter(ctx));
    if (!dom || !data) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (dom->suite != 1) return fail(DVN_ERR_INVALID_ARG, "domain is not a BLS12-377 domain");
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
    TRY(ntt_device<Fr377Cfg>(ctx, dom, d_data, inverse != 0, decimation, on_coset != 0));
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (!on_dev) HIP_TRY(hipMemcpyAsync(data, d_data, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipEventElapsedTime(&ctx->last_ms[0], ctx->ev[0], ctx->ev[1]));
    ctx->last_ms[1] = ctx->last_ms[0];
    return DVN_OK;
}
