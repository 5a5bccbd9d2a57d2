// BW6-761 instantiation: the curve of the aggregator proof (one per batch, 2^24-2^25
// constraints; /root/reference/spec/params/params.go:24, dispatch at /root/reference/prover/prover_gpu.go:36-43).
// G1 (y^2 = x^3 - 1) and G2 (y^2 = x^3 + 4) both live over the 761-bit base field: 27 limbs of 29 bits
// (R' = 2^783, external image 12 x u64 with R = 2^768); scalars are BLS12-377 base-field elements (377 bits,
// 6 x u64).  The group law never reads the curve constant, so one instantiation serves both groups.  With 27
// limbs the column accumulators of a product take 108 registers, so the field multiplication is a real call
// (Ops<Fe<C>>::CALL_MUL) and the lazily reduced accumulator formulas are off (three products per column would
// overflow 64 bits).
#include "suite_api.h"

static int32_t check_group(int group) { return (group == 1 || group == 2) ? DVN_OK : fail(DVN_ERR_INVALID_ARG, "group must be 1 (G1) or 2 (G2)"); }

extern "C" int32_t dvn_bw6761_points_upload(dvn_ctx* ctx, int group, const uint64_t* xy, size_t n, dvn_points** out) {
    TRY(check_group(group));
    return points_upload<G_761>(ctx, xy, n, group, out);
}
// P_i = [s0]B + i*[s1]B with B = `base_xy` (one affine point, gnark image) and (s0, s1) the first two 256-bit
// blocks of SplitMix64 stream `seed` (oracle/curves_py.py Suite.progression_points)
extern "C" int32_t dvn_bw6761_points_synthetic(dvn_ctx* ctx, int group, const uint64_t base_xy[24], uint64_t seed, size_t n, dvn_points** out) {
    TRY(enter(ctx));
    TRY(check_group(group));
    if (!out || !base_xy) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (n >= (1ull << 31)) return fail(DVN_ERR_INVALID_ARG, "point set too large");
    std::lock_guard<std::mutex> lk(ctx->mu);
    u32 s0[8], s1[8];
    for (int k = 0; k < 4; ++k) {
        u64 v = splitmix_at(seed, k); s0[2 * k] = (u32)v; s0[2 * k + 1] = (u32)(v >> 32);
        v = splitmix_at(seed, 4 + k); s1[2 * k] = (u32)v; s1[2 * k + 1] = (u32)(v >> 32);
    }
    const G_761::A gen = G_761::affine_from_ext((const u32*)base_xy);
    if (G_761::is_inf(gen)) return fail(DVN_ERR_INVALID_ARG, "base point is the point at infinity");
    G_761::A base = G_761::to_affine(host_scalar_mul<G_761>(gen, s0)), step = G_761::to_affine(host_scalar_mul<G_761>(gen, s1));
    return points_progression<G_761>(ctx, n, group, base, step, out);
}
extern "C" int32_t dvn_bw6761_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) {
    TRY(enter(ctx));
    if (!pts || (!xy && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->suite != 2) return fail(DVN_ERR_INVALID_ARG, "point set is not BW6-761");
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return points_download<G_761>(ctx, pts, off, n, xy);
}
extern "C" int32_t dvn_bw6761_msm(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[36]) {
    if (!pts) { TRY(enter(ctx)); return fail(DVN_ERR_INVALID_ARG, "NULL argument"); }
    return msm_full<G_761>(ctx, pts, pts->curve, off, n, scalars, flags, out_jac);
}
extern "C" int32_t dvn_bw6761_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) {
    TRY(enter(ctx));
    if (!d_out && n) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (kind != 0 && kind != 1) return fail(DVN_ERR_INVALID_ARG, "kind must be 0 (uniform) or 1 (witness-like)");
    if (n == 0) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    k_scalars_synthetic<Fp377Cfg><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((u32*)d_out, n, seed, kind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DVN_OK;
}
// ---- the scalar field's NTT (377-bit elements: 6 x uint64, two-adicity 46, coset shift 15) and computeH ------------------
extern "C" int32_t dvn_bw6761_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) { return ntt_domain_create<Fp377Cfg>(ctx, log_n, out); }
extern "C" int32_t dvn_bw6761_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) {
    return suite_ntt<Fp377Cfg>(ctx, "BW6-761", dom, data, inverse, decimation, on_coset, flags);
}
extern "C" int32_t dvn_bw6761_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                                size_t len, uint64_t* h_out, uint32_t flags) {
    return compute_h_api<Fp377Cfg>(ctx, "BW6-761", dom, a, b, c, len, h_out, flags);
}
