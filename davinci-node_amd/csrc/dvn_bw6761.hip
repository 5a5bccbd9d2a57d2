// BW6-761 instantiation: the curve of the aggregator proof (one per batch, 2^24-2^25
// constraints; /root/reference/spec/params/params.go:24, dispatch at /root/reference/prover/prover_gpu.go:36-43).
// G1 (y^2 = x^3 - 1) and G2 (y^2 = x^3 + 4) both live over the 761-bit base field: 27 limbs of 29 bits
// (R' = 2^783, external image 12 x u64 with R = 2^768); scalars are BLS12-377 base-field elements (377 bits,
// 6 x u64).  The group law never reads the curve constant, so one instantiation serves both groups.  With 27
// limbs the column accumulators of a product take 108 registers, so the field multiplication is a real call
// (Ops<Fe<C>>::CALL_MUL) and the lazily reduced accumulator formulas are off (three products per column would
// overflow 64 bits).
#include "suite_api.h"
#include "groth16_impl.h"
#include "artifacts_impl.h"

static int32_t check_group(int group) { return (group == 1 || group == 2) ? DVN_OK : fail(DVN_ERR_INVALID_ARG, "group must be 1 (G1) or 2 (G2)"); }

extern "C" int32_t dvn_bw6761_points_upload(dvn_ctx* ctx, int group, const uint64_t* xy, size_t n, dvn_points** out) try {
    TRY(check_group(group));
    return points_upload<G_761>(ctx, xy, n, group, out);
} DVN_CATCH
// P_i = [s0]B + i*[s1]B with B = `base_xy` (one affine point, gnark image) and (s0, s1) the first two 256-bit
// blocks of SplitMix64 stream `seed` (oracle/curves_py.py Suite.progression_points)
extern "C" int32_t dvn_bw6761_points_synthetic(dvn_ctx* ctx, int group, const uint64_t base_xy[24], uint64_t seed, size_t n, dvn_points** out) try {
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
} DVN_CATCH
extern "C" int32_t dvn_bw6761_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) try {
    TRY(enter(ctx));
    if (!pts || (!xy && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->suite != 2) return fail(DVN_ERR_INVALID_ARG, "point set is not BW6-761");
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return points_download<G_761>(ctx, pts, off, n, xy);
} DVN_CATCH
extern "C" int32_t dvn_bw6761_msm(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[36]) try {
    if (!pts) { TRY(enter(ctx)); return fail(DVN_ERR_INVALID_ARG, "NULL argument"); }
    return msm_full<G_761>(ctx, pts, pts->curve, off, n, scalars, flags, out_jac);
} DVN_CATCH
extern "C" int32_t dvn_bw6761_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) try {
    TRY(enter(ctx));
    if (!d_out && n) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (kind != 0 && kind != 1) return fail(DVN_ERR_INVALID_ARG, "kind must be 0 (uniform) or 1 (witness-like)");
    if (n == 0) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    k_scalars_synthetic<Fp377Cfg><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((u32*)d_out, n, seed, kind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DVN_OK;
} DVN_CATCH
// ---- the scalar field's NTT (377-bit elements: 6 x uint64, two-adicity 46, coset shift 15) and computeH ------------------
extern "C" int32_t dvn_bw6761_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) try { return ntt_domain_create<Fp377Cfg>(ctx, log_n, out); } DVN_CATCH
extern "C" int32_t dvn_bw6761_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) try {
    return suite_ntt<Fp377Cfg>(ctx, "BW6-761", dom, data, inverse, decimation, on_coset, flags);
} DVN_CATCH
extern "C" int32_t dvn_bw6761_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                                size_t len, uint64_t* h_out, uint32_t flags) try {
    return compute_h_api<Fp377Cfg>(ctx, "BW6-761", dom, a, b, c, len, h_out, flags);
} DVN_CATCH

// ---- Groth16 on BW6-761 (the aggregator proof, one per batch; reference dispatch prover/prover_gpu.go:52-57) ------------------
// Generators of the r-torsion for the SYNTHETIC setup: [#E / r] * (smallest-x point) on y^2 = x^3 - 1 and on y^2 = x^3 + 4,
// the group orders from the CM trace (oracle/curves_py.py Bw6761.g1_gen / g2_gen; tests/test_groth16_curves.py re-derives
// them).  Not gnark's generators -- production keys arrive through dvn_bw6761_points_upload.
static G_761::A bw6_g1_generator() {
    static const u64 g[24] = {
        0x851a1c1166ae2603ULL, 0xeea72e25492571beULL, 0xa6f20c75f7feab02ULL, 0xeeea8a1b066c47a4ULL, 0x6469c9e9402afc4fULL, 0x4cd9755dfe95e5baULL,
        0x3a1be33544ef7b02ULL, 0x1e7fe0555536b119ULL, 0xfd947c7fee73da05ULL, 0xcffde31744f837a0ULL, 0xa7b583036935c864ULL, 0x009918c73dcc124aULL,
        0xa3c13889f58ed694ULL, 0xee5cb8acda0ea650ULL, 0xcea7c1807dfc6e02ULL, 0xdc672db2da00df5dULL, 0xcde53070a22b574dULL, 0x4f3dc271f97fe60bULL,
        0xd5a2d25bb5f4cc8cULL, 0xd68088f33580a980ULL, 0x760b0d826b393143ULL, 0xe54149ee3970878eULL, 0xc62f69887a0e0ed1ULL, 0x0030ceaa8cbab39dULL};
    return G_761::affine_from_ext((const u32*)g);
}
static G_761::A bw6_g2_generator() {
    static const u64 g[24] = {
        0x7a0fd858704768c5ULL, 0x6c5be1b80c64008fULL, 0xe1b3fb4bb1ac5852ULL, 0x2203b091f0fc31ddULL, 0x2ebf37c25b1eef12ULL, 0x49a1fad2457203d9ULL,
        0x162db0b97521fa5cULL, 0xa6a1485ad8b21680ULL, 0xf5d702e72eed36deULL, 0xbef2f058a748dc3dULL, 0xcd04921580de2d02ULL, 0x011facfed9fa57fcULL,
        0xe09e2c209f095996ULL, 0x4d01106af3817932ULL, 0x492aba5ca1eb141aULL, 0x00746773a4274e9dULL, 0xa2dc76b92ff2c456ULL, 0xf966c58bd85998d4ULL,
        0x70f7b232c5c972f6ULL, 0x8397f5dd1e347c05ULL, 0x4ecba5f3f2ade37fULL, 0x8d8ebfa5a0d9b076ULL, 0xab65924d7383d531ULL, 0x0012b9419bd0fd43ULL};
    return G_761::affine_from_ext((const u32*)g);
}
static const u32 R761_W[12] = {0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu, 0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
struct SuiteBw6761 {
    typedef Fp377Cfg FrC;
    typedef G_761 G1T;
    typedef G_761 G2T;
    typedef FastHost<G_761>::G H1;
    typedef FastHost<G_761>::G H2;
    static constexpr int SUITE = 2;
    static const u32* r_words() { return R761_W; }
    static G_761::A g1_gen() { return bw6_g1_generator(); }
    static G_761::A g2_gen() { return bw6_g2_generator(); }
    static int32_t domain_create(dvn_ctx* ctx, u32 log_n, dvn_domain** out) { return ntt_domain_create<Fp377Cfg>(ctx, log_n, out); }
    static int32_t compute_h(dvn_ctx* ctx, const dvn_domain* dom, u32* a, u32* b, u32* c) { return compute_h_device<Fp377Cfg>(ctx, dom, a, b, c); }
    static int32_t msm_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) {
        return msm_submit<G_761>(ctx, p, off, n, (const uint64_t*)s, DVN_MSM_SCALARS_ON_DEVICE, l);
    }
    static int32_t msm_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) { return msm_submit_g1(ctx, p, off, n, s, l); }
    static int32_t msm_share_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return msm_share_submit<G_761>(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_share_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return msm_share_submit<G_761>(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_collect_g1(dvn_ctx* ctx, MsmLane* l, H1::X* out) { return msm_collect<G_761>(ctx, *l, out); }
    static int32_t msm_collect_g2(dvn_ctx* ctx, MsmLane* l, H2::X* out) { return msm_collect<G_761>(ctx, *l, out); }
};
extern "C" int32_t dvn_bw6761_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out) try {
    return r1cs_synthetic<SuiteBw6761>(seed, n_constraints, n_public, out);
} DVN_CATCH
extern "C" int32_t dvn_bw6761_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out) try {
    return groth16_setup<SuiteBw6761>(ctx, cs, seed, out);
} DVN_CATCH
// witness: m x 6 uint64; r_mont / s_mont: 6 uint64 each; proof_out: Ar (24) | Bs (24) | Krs (24) uint64
extern "C" int32_t dvn_bw6761_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                            const uint64_t r_mont[6], const uint64_t s_mont[6], uint64_t proof_out[72]) try {
    return groth16_prove<SuiteBw6761>(ctx, pk, witness, flags, r_mont, s_mont, proof_out);
} DVN_CATCH

// gnark's own key arrays + proofs from the solver's output on this curve (see dvn_groth16_pk_import / dvn_groth16_prove_solved)
extern "C" int32_t dvn_bw6761_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* k, dvn_groth16_pk** out) try { return groth16_pk_import_api<SuiteBw6761>(ctx, k, out); } DVN_CATCH
extern "C" int32_t dvn_bw6761_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t len,
                                              const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) try {
    return groth16_prove_solved<SuiteBw6761>(ctx, pk, w, a, b, c, len, r_mont, s_mont, proof_out);
} DVN_CATCH
// rank's share of dvn_bw6761_groth16_prove_solved and the assembly of the gathered shares (see dvn_groth16_prove_solved_shard / _combine)
extern "C" int32_t dvn_bw6761_groth16_prove_solved_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t len,
                                                    uint32_t rank, uint32_t world, uint64_t* partial_out) try {
    return groth16_prove_solved_shard<SuiteBw6761>(ctx, pk, w, a, b, c, len, rank, world, partial_out);
} DVN_CATCH
extern "C" int32_t dvn_bw6761_groth16_prove_combine(const dvn_groth16_pk* pk, const uint64_t* partials, uint32_t world, const uint64_t* r_mont, const uint64_t* s_mont,
                                               uint64_t* proof_out) try {
    return groth16_prove_combine<SuiteBw6761>(pk, partials, world, r_mont, s_mont, proof_out);
} DVN_CATCH
extern "C" int32_t dvn_bw6761_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words) try {
    return groth16_pk_export_arrays<SuiteBw6761>(ctx, pk, which, out, cap_words, n_words);
} DVN_CATCH
// Pedersen commitment + PoK on this curve: the BSB22 commitment hint of the aggregator proof (gnark-crypto
// ecc/bw6-761/fr/pedersen ProvingKey.Commit / ProveKnowledge); values are 6-word scalars.
extern "C" int32_t dvn_bw6761_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma, const uint64_t* values_mont,
                                              size_t n, uint32_t flags, uint64_t commitment_out[24], uint64_t pok_out[24]) try {
    return pedersen_commit<SuiteBw6761>(ctx, basis, basis_exp_sigma, values_mont, n, flags, commitment_out, pok_out);
} DVN_CATCH
// The aggregator proof over all devices of a multi-device context (see dvn_groth16_prove_solved_multi)
extern "C" int32_t dvn_bw6761_groth16_prove_solved_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* w, const uint64_t* a,
                                                         const uint64_t* b, const uint64_t* c, size_t len, const uint64_t* r_mont, const uint64_t* s_mont,
                                                         uint64_t* proof_out) try {
    if (!a || !b || !c) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const uint64_t* src[3] = {a, b, c};
    return groth16_prove_multi<SuiteBw6761>(ctx, keys, nkeys, w, 0u, src, len, r_mont, s_mont, proof_out);
} DVN_CATCH


// ---- the aggregator's proving key as the node stores it (gnark pk.WriteTo, /root/reference/cmd/circuit-compile/main.go:507-513;
// loaded at boot for the aggregator circuit, /root/reference/sequencer/circuit_artifacts.go:39-76): artifacts_impl.h on this
// curve -- 96-byte coordinates, three flag bits, a^((p + 1) / 4) square roots, both groups curves over Fp ------------------------
namespace dvn_art {
template <> struct KeyTraits<SuiteBw6761> {
    static constexpr const char* NAME = "BW6-761";
    static constexpr u32 COSET_GEN = 15;                                  // fr.MultiplicativeGen of BW6-761
    static CurveAux1<Fp761Cfg> aux_g1() { return make_aux1<Fp761Cfg>(Ops<Fp761>::neg(fe_one<Fp761Cfg>()), 0); }                          // y^2 = x^3 - 1
    static CurveAux1<Fp761Cfg> aux_g2() { return make_aux1<Fp761Cfg>(fe_from_canonical<Fp761Cfg>(fe_small<Fp761Cfg>(4)), 0); }            // y^2 = x^3 + 4
};
}  // namespace dvn_art
extern "C" int32_t dvn_bw6761_gnark_pk_scan(const uint8_t* data, size_t len, uint64_t info[12]) try { return dvn_art::pk_scan_api<SuiteBw6761>(data, len, info); } DVN_CATCH
extern "C" int32_t dvn_bw6761_groth16_pk_load_bytes(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                                    uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    return dvn_art::pk_load_bytes_api<SuiteBw6761>(ctx, data, len, n_public, committed_wires, n_committed_wires, out);
} DVN_CATCH
extern "C" int32_t dvn_bw6761_groth16_pk_load_file(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                                   const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    return dvn_art::pk_load_file_api<SuiteBw6761>(ctx, path, sha256_hex, n_public, committed_wires, n_committed_wires, out);
} DVN_CATCH
