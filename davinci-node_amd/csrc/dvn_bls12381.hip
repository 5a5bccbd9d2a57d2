// BLS12-381 instantiation: the fourth curve of the reference's GPU dispatch (/root/reference/prover/prover_gpu.go:45-50)
// and the curve of the EIP-4844 blob commitment that precedes every state-transition proof (a
// 4096-point G1 MSM over the KZG setup, /root/reference/crypto/blobs/blob.go:40-90, /root/reference/types/blobs.go:90-160).
// Same kernels: 14-limb base field (381 bits, external image 6 x u64, R = 2^384), Fe2 with u^2 = -1 for G2, and the
// 255-bit scalar field (two-adicity 32, coset shift 7) for the NTT.
#include "suite_api.h"
#include "groth16_impl.h"

static G1_381::A g1_381_generator() {      // the standard BLS12-381 G1 generator, Montgomery image (oracle/curves_py.py BLS12_381.g1_gen)
    static const u64 g[12] = {0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL, 0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL,
                              0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL, 0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL};
    return G1_381::affine_from_ext((const u32*)g);
}
static const u32 R381_W[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};

extern "C" int32_t dvn_bls12381_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) try { return points_upload<G1_381>(ctx, xy, n, 1, out); } DVN_CATCH
extern "C" int32_t dvn_bls12381_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) try { return points_upload<G2_381>(ctx, xy, n, 2, out); } DVN_CATCH
extern "C" int32_t dvn_bls12381_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out) try {
    return suite_points_synthetic<G1_381>(ctx, 1, g1_381_generator(), R381_W, seed, n, out);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) try {
    return suite_points_download<G1_381, G2_381>(ctx, "BLS12-381", pts, off, n, xy);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[18]) try {
    return msm_full<G1_381>(ctx, pts, 1, off, n, scalars, flags, out_jac);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[36]) try {
    return msm_full<G2_381>(ctx, pts, 2, off, n, scalars, flags, out_jac);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) try { return suite_scalars_synthetic<Fr381Cfg>(ctx, seed, kind, n, d_out); } DVN_CATCH
extern "C" int32_t dvn_bls12381_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) try { return ntt_domain_create<Fr381Cfg>(ctx, log_n, out); } DVN_CATCH
extern "C" int32_t dvn_bls12381_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) try {
    return suite_ntt<Fr381Cfg>(ctx, "BLS12-381", dom, data, inverse, decimation, on_coset, flags);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                                 size_t len, uint64_t* h_out, uint32_t flags) try {
    return compute_h_api<Fr381Cfg>(ctx, "BLS12-381", dom, a, b, c, len, h_out, flags);
} DVN_CATCH

// ---- Groth16 on BLS12-381 (fourth curve of the reference's GPU dispatch, prover/prover_gpu.go:45-50) -------------------------
static G2_381::A g2_381_generator() {      // the standard BLS12-381 G2 generator, Montgomery image (oracle/curves_py.py BLS12_381.g2_gen)
    static const u64 g[24] = {
        0xf5f28fa202940a10ULL, 0xb3f5fb2687b4961aULL, 0xa1a893b53e2ae580ULL, 0x9894999d1a3caee9ULL, 0x6f67b7631863366bULL, 0x058191924350bcd7ULL,
        0xa5a9c0759e23f606ULL, 0xaaa0c59dbccd60c3ULL, 0x3bb17e18e2867806ULL, 0x1b1ab6cc8541b367ULL, 0xc2b6ed0ef2158547ULL, 0x11922a097360edf3ULL,
        0x4c730af860494c4aULL, 0x597cfa1f5e369c5aULL, 0xe7e6856caa0a635aULL, 0xbbefb5e96e0d495fULL, 0x07d3a975f0ef25a2ULL, 0x0083fd8e7e80dae5ULL,
        0xadc0fc92df64b05dULL, 0x18aa270a2b1461dcULL, 0x86adac6a3be4eba0ULL, 0x79495c4ec93da33aULL, 0xe7175850a43ccaedULL, 0x0b2bc2a163de1bf2ULL};
    return G2_381::affine_from_ext((const u32*)g);
}
struct SuiteBls12381 {
    typedef Fr381Cfg FrC;
    typedef G1_381 G1T;
    typedef G2_381 G2T;
    typedef FastHost<G1_381>::G H1;
    typedef FastHost<G2_381>::G H2;
    static constexpr int SUITE = 3;
    static const u32* r_words() { return R381_W; }
    static G1_381::A g1_gen() { return g1_381_generator(); }
    static G2_381::A g2_gen() { return g2_381_generator(); }
    static int32_t domain_create(dvn_ctx* ctx, u32 log_n, dvn_domain** out) { return ntt_domain_create<Fr381Cfg>(ctx, log_n, out); }
    static int32_t compute_h(dvn_ctx* ctx, const dvn_domain* dom, u32* a, u32* b, u32* c) { return compute_h_device<Fr381Cfg>(ctx, dom, a, b, c); }
    static int32_t msm_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) {
        return msm_submit<G1_381>(ctx, p, off, n, (const uint64_t*)s, DVN_MSM_SCALARS_ON_DEVICE, l);
    }
    static int32_t msm_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) {
        return msm_submit<G2_381>(ctx, p, off, n, (const uint64_t*)s, DVN_MSM_SCALARS_ON_DEVICE, l);
    }
    static int32_t msm_share_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return msm_share_submit<G1_381>(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_share_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return msm_share_submit<G2_381>(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_collect_g1(dvn_ctx* ctx, MsmLane* l, H1::X* out) { return msm_collect<G1_381>(ctx, *l, out); }
    static int32_t msm_collect_g2(dvn_ctx* ctx, MsmLane* l, H2::X* out) { return msm_collect<G2_381>(ctx, *l, out); }
};
extern "C" int32_t dvn_bls12381_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out) try {
    return r1cs_synthetic<SuiteBls12381>(seed, n_constraints, n_public, out);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out) try {
    return groth16_setup<SuiteBls12381>(ctx, cs, seed, out);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                              const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[48]) try {
    return groth16_prove<SuiteBls12381>(ctx, pk, witness, flags, r_mont, s_mont, proof_out);
} DVN_CATCH

// gnark's own key arrays + proofs from the solver's output on this curve (see dvn_groth16_pk_import / dvn_groth16_prove_solved)
extern "C" int32_t dvn_bls12381_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* k, dvn_groth16_pk** out) try { return groth16_pk_import_api<SuiteBls12381>(ctx, k, out); } DVN_CATCH
extern "C" int32_t dvn_bls12381_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t len,
                                              const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) try {
    return groth16_prove_solved<SuiteBls12381>(ctx, pk, w, a, b, c, len, r_mont, s_mont, proof_out);
} DVN_CATCH
extern "C" int32_t dvn_bls12381_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words) try {
    return groth16_pk_export_arrays<SuiteBls12381>(ctx, pk, which, out, cap_words, n_words);
} DVN_CATCH

// ---- blob KZG (EIP-4844 / EIP-7594) on this curve ----------------------------------------------------
#include "kzg_impl.h"
// Pedersen commitment + PoK on this curve (gnark-crypto ecc/bls12-381/fr/pedersen ProvingKey.Commit / ProveKnowledge)
extern "C" int32_t dvn_bls12381_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma, const uint64_t* values_mont,
                                                size_t n, uint32_t flags, uint64_t commitment_out[12], uint64_t pok_out[12]) try {
    return pedersen_commit<SuiteBls12381>(ctx, basis, basis_exp_sigma, values_mont, n, flags, commitment_out, pok_out);
} DVN_CATCH
