// BLS12-377 instantiation: the curve of the voteverifier proofs (60 per batch,
// /root/reference/spec/params/params.go:21-27; GPU dispatch at /root/reference/prover/prover_gpu.go:36-43).
// The same kernels as BN254, instantiated on 14-limb base-field elements (377 bits, external image 6 x u64,
// R = 2^384), Fe2 with u^2 = -5 for G2, and the 253-bit scalar field (two-adicity 47, coset shift 22) for the NTT.
#include "suite_api.h"
#include "groth16_impl.h"
#include "artifacts_impl.h"

static G1_377::A g1_377_generator() {      // gnark-crypto bls12-377 G1 generator, Montgomery image (oracle/curves_py.py BLS12_377.g1_gen)
    static const u64 g[12] = {0x260f33b9772451f4ULL, 0xc54dd773169d5658ULL, 0x5c1551c469a510ddULL, 0x761662e4425e1698ULL, 0xc97d78cc6f065272ULL, 0x00a41206b361fd4dULL,
                              0x8193961fb8cb81f3ULL, 0x00638d4c5f44adb8ULL, 0xfafaf3dad4daf54aULL, 0xc27849e2d655cd18ULL, 0x2ec3ddb401d52814ULL, 0x007da93326303c71ULL};
    return G1_377::affine_from_ext((const u32*)g);
}
static const u32 R377_W[8] = {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};

extern "C" int32_t dvn_bls12377_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) try { return points_upload<G1_377>(ctx, xy, n, 1, out); } DVN_CATCH
extern "C" int32_t dvn_bls12377_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) try { return points_upload<G2_377>(ctx, xy, n, 2, out); } DVN_CATCH
extern "C" int32_t dvn_bls12377_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out) try {
    return suite_points_synthetic<G1_377>(ctx, 1, g1_377_generator(), R377_W, seed, n, out);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) try {
    return suite_points_download<G1_377, G2_377>(ctx, "BLS12-377", pts, off, n, xy);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[18]) try {
    return msm_full<G1_377>(ctx, pts, 1, off, n, scalars, flags, out_jac);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[36]) try {
    return msm_full<G2_377>(ctx, pts, 2, off, n, scalars, flags, out_jac);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) try { return suite_scalars_synthetic<Fr377Cfg>(ctx, seed, kind, n, d_out); } DVN_CATCH
extern "C" int32_t dvn_bls12377_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) try { return ntt_domain_create<Fr377Cfg>(ctx, log_n, out); } DVN_CATCH
extern "C" int32_t dvn_bls12377_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) try {
    return suite_ntt<Fr377Cfg>(ctx, "BLS12-377", dom, data, inverse, decimation, on_coset, flags);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                                 size_t len, uint64_t* h_out, uint32_t flags) try {
    return compute_h_api<Fr377Cfg>(ctx, "BLS12-377", dom, a, b, c, len, h_out, flags);
} DVN_CATCH

// ---- Groth16 on BLS12-377 (the voteverifier proof, 60 per batch): groth16_impl.h over this curve's MSMs and computeH -----------
// Synthetic Setup needs a generator of the r-torsion of the twist: [#E'(Fp2) / r] * Q for the smallest-x twist point Q
// (oracle/curves_py.py BLS12_377.g2_gen; NOT gnark's generator -- production keys arrive through points_upload).
static G2_377::A g2_377_generator() {
    static const u64 g[24] = {
        0x87e4cb96756a6edbULL, 0x99aa13a4fa9b9c4bULL, 0xd702cf5f38b86463ULL, 0x7ed4ce7924c167edULL, 0x1b447e91a9c5329fULL, 0x0084963f63a498b7ULL,
        0x8554a79176a27e27ULL, 0xba4edb07f79e64a3ULL, 0xe2dc99bdfb15b037ULL, 0xe7d8740ff9d797f7ULL, 0x140c90e86aa2ea12ULL, 0x00cab5f295c622d3ULL,
        0x2050bf7247d3ced4ULL, 0x67a8ee8100d17921ULL, 0x17d6f6cda3578307ULL, 0xd0eaa7550e0e26d8ULL, 0xf7aa721a161d7284ULL, 0x01604cec93c28ee1ULL,
        0x4fc1f83c331faed0ULL, 0x6783c3d79d164e16ULL, 0x7ecc86be7de427ffULL, 0xbe7b613ac02526d5ULL, 0xb5eb2e9573178ceeULL, 0x00f5f8619a031dbfULL};
    return G2_377::affine_from_ext((const u32*)g);
}
struct SuiteBls12377 {
    typedef Fr377Cfg FrC;
    typedef G1_377 G1T;
    typedef G2_377 G2T;
    typedef FastHost<G1_377>::G H1;
    typedef FastHost<G2_377>::G H2;
    static constexpr int SUITE = 1;
    static const u32* r_words() { return R377_W; }
    static G1_377::A g1_gen() { return g1_377_generator(); }
    static G2_377::A g2_gen() { return g2_377_generator(); }
    static int32_t domain_create(dvn_ctx* ctx, u32 log_n, dvn_domain** out) { return ntt_domain_create<Fr377Cfg>(ctx, log_n, out); }
    static int32_t compute_h(dvn_ctx* ctx, const dvn_domain* dom, u32* a, u32* b, u32* c) { return compute_h_device<Fr377Cfg>(ctx, dom, a, b, c); }
    static int32_t msm_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) {
        return msm_submit<G1_377>(ctx, p, off, n, (const uint64_t*)s, DVN_MSM_SCALARS_ON_DEVICE, l);
    }
    static int32_t msm_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) {
        return msm_submit<G2_377>(ctx, p, off, n, (const uint64_t*)s, DVN_MSM_SCALARS_ON_DEVICE, l);
    }
    static int32_t msm_share_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return msm_share_submit<G1_377>(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_share_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return msm_share_submit<G2_377>(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_collect_g1(dvn_ctx* ctx, MsmLane* l, H1::X* out) { return msm_collect<G1_377>(ctx, *l, out); }
    static int32_t msm_collect_g2(dvn_ctx* ctx, MsmLane* l, H2::X* out) { return msm_collect<G2_377>(ctx, *l, out); }
};
extern "C" int32_t dvn_bls12377_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out) try {
    return r1cs_synthetic<SuiteBls12377>(seed, n_constraints, n_public, out);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out) try {
    return groth16_setup<SuiteBls12377>(ctx, cs, seed, out);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                              const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[48]) try {
    return groth16_prove<SuiteBls12377>(ctx, pk, witness, flags, r_mont, s_mont, proof_out);
} DVN_CATCH

// gnark's own key arrays + proofs from the solver's output on this curve (see dvn_groth16_pk_import / dvn_groth16_prove_solved)
extern "C" int32_t dvn_bls12377_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* k, dvn_groth16_pk** out) try { return groth16_pk_import_api<SuiteBls12377>(ctx, k, out); } DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t len,
                                              const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) try {
    return groth16_prove_solved<SuiteBls12377>(ctx, pk, w, a, b, c, len, r_mont, s_mont, proof_out);
} DVN_CATCH
// rank's share of dvn_bls12377_groth16_prove_solved and the assembly of the gathered shares (see dvn_groth16_prove_solved_shard / _combine)
extern "C" int32_t dvn_bls12377_groth16_prove_solved_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b, const uint64_t* c, size_t len,
                                                    uint32_t rank, uint32_t world, uint64_t* partial_out) try {
    return groth16_prove_solved_shard<SuiteBls12377>(ctx, pk, w, a, b, c, len, rank, world, partial_out);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_prove_combine(const dvn_groth16_pk* pk, const uint64_t* partials, uint32_t world, const uint64_t* r_mont, const uint64_t* s_mont,
                                               uint64_t* proof_out) try {
    return groth16_prove_combine<SuiteBls12377>(pk, partials, world, r_mont, s_mont, proof_out);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words) try {
    return groth16_pk_export_arrays<SuiteBls12377>(ctx, pk, which, out, cap_words, n_words);
} DVN_CATCH
// Pedersen commitment + PoK on this curve: the BSB22 commitment hint of the voteverifier proofs (gnark-crypto
// ecc/bls12-377/fr/pedersen ProvingKey.Commit / ProveKnowledge); the hash-to-field of the challenge stays in gnark.
extern "C" int32_t dvn_bls12377_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma, const uint64_t* values_mont,
                                                size_t n, uint32_t flags, uint64_t commitment_out[12], uint64_t pok_out[12]) try {
    return pedersen_commit<SuiteBls12377>(ctx, basis, basis_exp_sigma, values_mont, n, flags, commitment_out, pok_out);
} DVN_CATCH


// ---- the voteverifier's proving key as the node stores it (gnark pk.WriteTo, /root/reference/cmd/circuit-compile/main.go:507-513;
// loaded at boot for the voteverifier circuit, /root/reference/sequencer/circuit_artifacts.go:39-76): artifacts_impl.h on this
// curve -- 48-byte coordinates, three flag bits, Tonelli-Shanks square roots (p = 1 mod 4, two-adicity 46) ------------------------
namespace dvn_art {
template <> struct KeyTraits<SuiteBls12377> {
    static constexpr const char* NAME = "BLS12-377";
    static constexpr u32 COSET_GEN = 22;                                  // fr.MultiplicativeGen of BLS12-377
    static constexpr u32 NON_RESIDUE = 5;                                 // u^2 = -5 is irreducible and -1 is a square (p = 1 mod 4)
    static CurveAux1<Fp377Cfg> aux_g1() { return make_aux1<Fp377Cfg>(fe_one<Fp377Cfg>(), NON_RESIDUE); }               // y^2 = x^3 + 1
    static CurveAux2<Fp377Cfg> aux_g2() {                                                                           // twist: y^2 = x^3 + 1/u = x^3 - u/5
        typedef Ops<Fp377> O;
        const Fp377 fifth = O::inv(fe_from_canonical<Fp377Cfg>(fe_small<Fp377Cfg>(5)));
        return make_aux2<Fp377Cfg>(Fp2_377{fe_zero<Fp377Cfg>(), O::neg(fifth)}, NON_RESIDUE);
    }
};
}  // namespace dvn_art
extern "C" int32_t dvn_bls12377_gnark_pk_scan(const uint8_t* data, size_t len, uint64_t info[12]) try { return dvn_art::pk_scan_api<SuiteBls12377>(data, len, info); } DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_pk_load_bytes(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                                      uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    return dvn_art::pk_load_bytes_api<SuiteBls12377>(ctx, data, len, n_public, committed_wires, n_committed_wires, out);
} DVN_CATCH
extern "C" int32_t dvn_bls12377_groth16_pk_load_file(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                                     const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    return dvn_art::pk_load_file_api<SuiteBls12377>(ctx, path, sha256_hex, n_public, committed_wires, n_committed_wires, out);
} DVN_CATCH
