// Groth16 on BN254 (the statetransition curve): the C-ABI entry points over the curve-generic orchestration of
// groth16_impl.h, plus what only exists for this curve -- the Solidity ABI image of a proof and the Pedersen
// commitment of gnark's BSB22 hint.  The MSMs and computeH of this suite are the ones compiled in dvn_msm_g1.hip,
// dvn_msm_g2.hip and dvn_ntt.hip (cross-TU calls: the kernels are not instantiated a second time here).
#include "groth16_impl.h"
#include "suite_bn254.h"


extern "C" int32_t dvn_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out) try {
    return r1cs_synthetic<SuiteBn254>(seed, n_constraints, n_public, out);
} DVN_CATCH
extern "C" int32_t dvn_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out) try { return groth16_setup<SuiteBn254>(ctx, cs, seed, out); } DVN_CATCH
extern "C" int32_t dvn_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                     const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]) try {
    return groth16_prove<SuiteBn254>(ctx, pk, witness, flags, r_mont, s_mont, proof_out);
} DVN_CATCH

// Sharded proving (BASELINE config 5: the proof's MSMs sharded across the GPUs of a node): rank `rank` of `world` computes its
// rows of the five MSMs (dvn_groth16_prove_shard, 72 uint64 of partial results), the ranks all-gather them, and every rank (or
// one) assembles the proof on the host (dvn_groth16_prove_combine).  Key sets too small to be expanded (< 2^10 points) are computed whole by rank 0.
extern "C" int32_t dvn_groth16_prove_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                           uint32_t rank, uint32_t world, uint64_t partial_out[72]) try {
    return groth16_prove_shard<SuiteBn254>(ctx, pk, witness, flags, rank, world, partial_out);
} DVN_CATCH
extern "C" int32_t dvn_groth16_prove_solved_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                                  size_t len, uint32_t rank, uint32_t world, uint64_t partial_out[72]) try {
    return groth16_prove_solved_shard<SuiteBn254>(ctx, pk, w, a, b, c, len, rank, world, partial_out);
} DVN_CATCH
extern "C" int32_t dvn_groth16_prove_combine(const dvn_groth16_pk* pk, const uint64_t* partials, uint32_t world,
                                             const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]) try {
    return groth16_prove_combine<SuiteBn254>(pk, partials, world, r_mont, s_mont, proof_out);
} DVN_CATCH

// One proof over all devices of a multi-device context (dvn_init(devs, ndev > 1)); keys[i] = the key on dvn_ctx_device(ctx, i)
extern "C" int32_t dvn_groth16_prove_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* witness,
                                           const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]) try {
    return groth16_prove_multi<SuiteBn254>(ctx, keys, nkeys, witness, 0u, nullptr, 0, r_mont, s_mont, proof_out);
} DVN_CATCH
extern "C" int32_t dvn_groth16_prove_solved_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* w, const uint64_t* a,
                                                  const uint64_t* b, const uint64_t* c, size_t len, const uint64_t r_mont[4], const uint64_t s_mont[4],
                                                  uint64_t proof_out[32]) try {
    if (!a || !b || !c) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const uint64_t* src[3] = {a, b, c};
    return groth16_prove_multi<SuiteBn254>(ctx, keys, nkeys, w, 0u, src, len, r_mont, s_mont, proof_out);
} DVN_CATCH
extern "C" int32_t dvn_groth16_last_stats_json(const dvn_groth16_pk* pk, char* buf, size_t cap, size_t* needed) try {
    if (!pk) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    return groth16_stats_json(pk, buf, cap, needed);
} DVN_CATCH

// ---- curve-independent handles ------------------------------------------------------------------------------------
// gnark's ecc.ID numbering of the prover package mirror: 1 BN254, 2 BLS12_377, 3 BLS12_381, 4 BW6_761
extern "C" int32_t dvn_r1cs_curve(const dvn_r1cs* cs) try { return !cs ? 0 : cs->suite == 0 ? 1 : cs->suite == 1 ? 2 : cs->suite == 3 ? 3 : cs->suite == 2 ? 4 : 0; } DVN_CATCH
extern "C" int32_t dvn_r1cs_info(const dvn_r1cs* cs, uint32_t* n_wires, uint32_t* n_public, uint32_t* n_constraints) try {
    if (!cs) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (n_wires) *n_wires = cs->n_wires;
    if (n_public) *n_public = cs->n_public;
    if (n_constraints) *n_constraints = cs->n_constraints;
    return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_r1cs_witness(const dvn_r1cs* cs, uint64_t* out) try {
    if (!cs || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    memcpy(out, cs->witness.data(), cs->witness.size() * 4);
    return DVN_OK;
} DVN_CATCH
extern "C" void dvn_r1cs_free(dvn_r1cs* cs) { delete cs; }

extern "C" int32_t dvn_groth16_pk_free(dvn_ctx* ctx, dvn_groth16_pk* pk) try {
    TRY(enter(ctx));
    if (!pk) return DVN_OK;
    if (pk->cached) return DVN_OK;             // resident in the artifact cache until dvn_shutdown
    dvn_points* ps[] = {pk->A, pk->B, pk->K, pk->Z, pk->B2, pk->Basis, pk->BasisSigma};
    for (dvn_points* p : ps) if (p) (void)dvn_points_free(ctx, p);
    if (pk->dom) (void)dvn_ntt_domain_free(ctx, pk->dom);
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipStreamSynchronize(ctx->stream);
    u32* ptrs[] = {pk->d_idxK, pk->d_wK, pk->d_idxA, pk->d_idxB, pk->d_ptr[0], pk->d_ptr[1], pk->d_ptr[2], pk->d_idx[0], pk->d_idx[1], pk->d_idx[2],
                   pk->d_coef[0], pk->d_coef[1], pk->d_coef[2], pk->d_w, pk->d_abc, pk->d_wA, pk->d_wB};
    for (u32* p : ptrs) if (p) (void)hipFree(p);
    if (pk->hk && pk->hk_free) pk->hk_free(pk->hk);
    delete pk;
    return DVN_OK;
} DVN_CATCH

extern "C" int32_t dvn_groth16_pk_info(const dvn_groth16_pk* pk, uint32_t* log_n, uint64_t* sizes /* nA, nB, nK, nZ, nB2 */) try {
    if (!pk) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (log_n) *log_n = pk->log_n;
    if (sizes) { sizes[0] = pk->nA; sizes[1] = pk->nB; sizes[2] = pk->m - pk->k_first; sizes[3] = pk->n; sizes[4] = pk->nB; }
    return DVN_OK;
} DVN_CATCH

// ---- gnark's own proving key, handed over once; then proofs from the solver's output -------------------------------------------
extern "C" int32_t dvn_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* k, dvn_groth16_pk** out) try { return groth16_pk_import_api<SuiteBn254>(ctx, k, out); } DVN_CATCH
extern "C" int32_t dvn_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                            const uint64_t* c, size_t len, const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]) try {
    return groth16_prove_solved<SuiteBn254>(ctx, pk, w, a, b, c, len, r_mont, s_mont, proof_out);
} DVN_CATCH
extern "C" int32_t dvn_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words) try {
    return groth16_pk_export_arrays<SuiteBn254>(ctx, pk, which, out, cap_words, n_words);
} DVN_CATCH

// The verifying key Setup produced next to the proving key (groth16.Setup returns both, /root/reference/prover/setup.go:15-28),
// as gnark affine Montgomery images: alpha (G1) | beta | gamma | delta (G2) | IC[n_public] (G1) | with a commitment: Pedersen
// G | GSigmaNeg (G2).  n_words receives the total; out may be NULL to query it.
extern "C" int32_t dvn_groth16_vk_export(const dvn_groth16_pk* pk, uint64_t* out, size_t cap_words, size_t* n_words) try {
    if (!pk || !n_words) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    *n_words = pk->vk_words.size();
    if (!out) return DVN_OK;
    if (cap_words < pk->vk_words.size()) return fail(DVN_ERR_INVALID_ARG, "buffer holds %zu words, the key needs %zu", cap_words, pk->vk_words.size());
    memcpy(out, pk->vk_words.data(), pk->vk_words.size() * 8);
    return DVN_OK;
} DVN_CATCH

extern "C" int32_t dvn_groth16_last_ms(const dvn_groth16_pk* pk, float* ms_h, float* ms_msm, float* ms_total) try {
    if (!pk) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (ms_h) *ms_h = pk->ms_h;
    if (ms_msm) *ms_msm = pk->ms_msm;
    if (ms_total) *ms_total = pk->ms_total;
    return DVN_OK;
} DVN_CATCH


// ------------------------------------------------------------------------------------------------------
// Solidity ABI image of a proof: /root/reference/solidity/solidity.go:29-73,85-116
// (uint256[8] proof, uint256[2] commitments, uint256[2] commitmentPok), every word big-endian, G2 coordinates in
// EIP-197 order (A1 before A0).  Host only.
// ------------------------------------------------------------------------------------------------------
static void fp_ext_to_be32(const u32* ext_words, uint8_t* out) {
    u32 t[8], c[8];
    memcpy(t, ext_words, 32);
    fe_pack<FpCfg>(fe_canonical<FpCfg>(fe_mul<FpCfg>(fe_unpack<FpCfg>(t), fe_small<FpCfg>(32))), c);   // x*2^256 -> x
    for (int i = 0; i < 8; ++i) {
        const u32 w = c[7 - i];
        out[4 * i] = (uint8_t)(w >> 24); out[4 * i + 1] = (uint8_t)(w >> 16); out[4 * i + 2] = (uint8_t)(w >> 8); out[4 * i + 3] = (uint8_t)w;
    }
}
extern "C" int32_t dvn_groth16_proof_abi_encode(const uint64_t proof[32], const uint64_t commitment[8], const uint64_t commitment_pok[8],
                                                uint8_t out[384]) try {
    if (!proof || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const u32* p = (const u32*)proof;
    const u32* ar = p; const u32* bs = p + 16; const u32* krs = p + 48;
    fp_ext_to_be32(ar, out); fp_ext_to_be32(ar + 8, out + 32);
    fp_ext_to_be32(bs + 8, out + 64);  fp_ext_to_be32(bs, out + 96);          // Bs.X.A1, Bs.X.A0
    fp_ext_to_be32(bs + 24, out + 128); fp_ext_to_be32(bs + 16, out + 160);   // Bs.Y.A1, Bs.Y.A0
    fp_ext_to_be32(krs, out + 192); fp_ext_to_be32(krs + 8, out + 224);
    memset(out + 256, 0, 128);
    if (commitment) { fp_ext_to_be32((const u32*)commitment, out + 256); fp_ext_to_be32((const u32*)commitment + 8, out + 288); }
    if (commitment_pok) { fp_ext_to_be32((const u32*)commitment_pok, out + 320); fp_ext_to_be32((const u32*)commitment_pok + 8, out + 352); }
    return DVN_OK;
} DVN_CATCH


// ------------------------------------------------------------------------------------------------------
// Pedersen commitment + proof of knowledge (gnark-crypto ecc/bn254/fr/pedersen ProvingKey.Commit /
// ProveKnowledge, called from gnark's BSB22 commitment hint inside groth16.Prove): two G1 MSMs over the committed
// wire values, against Basis and BasisExpSigma.  With a single commitment (statetransition has exactly one,
// /root/reference/solidity/solidity.go:57-65) BatchProve's fold is the identity, so pok_out IS CommitmentPok.
// Outputs are G1Affine images (8 x uint64, infinity = zeros).  Both MSMs are in flight together (two lanes).
// ------------------------------------------------------------------------------------------------------
extern "C" int32_t dvn_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma,
                                       const uint64_t* values_mont, size_t n, uint32_t flags,
                                       uint64_t commitment_out[8], uint64_t pok_out[8]) try {
    return pedersen_commit<SuiteBn254>(ctx, basis, basis_exp_sigma, values_mont, n, flags, commitment_out, pok_out);
} DVN_CATCH


// ------------------------------------------------------------------------------------------------------
// Groth16 with ONE BSB22 commitment -- the statetransition shape: 8 public inputs + 1 commitment wire
// (/root/reference/config/statetransition_vkey.sol:97-114, solidity/solidity.go:57-65).  gnark's Prove
// computes, inside the solver's commitment hint, Commitment = MSM(pk.CommitmentKeys[0].Basis, committed wires),
// derives the challenge that becomes the commitment wire, finishes the solve, and adds CommitmentPok =
// MSM(BasisExpSigma, committed wires) (one commitment: BatchProve's fold is the identity).  The challenge is what the
// reference's Solidity verifier recomputes: keccak256(C.x || C.y || publicCommitted) mod r (statetransition_vkey.sol:659-676).
// ------------------------------------------------------------------------------------------------------
static void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
                                    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
                                    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
                                    0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   // [x + 5y]
    auto rol = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) st[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(st[x + 5 * y], ROT[x + 5 * y]);
        for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) st[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        st[0] ^= RC[round];
    }
}
// legacy Keccak-256 (Ethereum / Solidity keccak256): rate 136, padding 0x01 .. 0x80
static void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint64_t st[25] = {0};
    std::vector<uint8_t> p(data, data + len);
    p.push_back(0x01);
    while (p.size() % 136) p.push_back(0);
    p.back() |= 0x80;
    for (size_t off = 0; off < p.size(); off += 136) {
        for (int i = 0; i < 17; ++i) { uint64_t v = 0; for (int k = 7; k >= 0; --k) v = (v << 8) | p[off + 8 * i + k]; st[i] ^= v; }
        keccak_f1600(st);
    }
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 8; ++k) out[8 * i + k] = (uint8_t)(st[i] >> (8 * k));
}
extern "C" int32_t dvn_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) try {
    if ((!data && len) || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    keccak256(data, len, out);
    return DVN_OK;
} DVN_CATCH
static void fr_ext_to_be32(const u32* ext_words, uint8_t* out) {
    u32 t[8], c[8];
    memcpy(t, ext_words, 32);
    fe_pack<FrCfg>(fe_canonical<FrCfg>(fe_mul<FrCfg>(fe_unpack<FrCfg>(t), fe_small<FrCfg>(32))), c);   // x*2^256 -> x
    for (int i = 0; i < 8; ++i) {
        const u32 w = c[7 - i];
        out[4 * i] = (uint8_t)(w >> 24); out[4 * i + 1] = (uint8_t)(w >> 16); out[4 * i + 2] = (uint8_t)(w >> 8); out[4 * i + 3] = (uint8_t)w;
    }
}

extern "C" int32_t dvn_r1cs_synthetic_committed(uint64_t seed, uint32_t n_constraints, uint32_t n_public, uint32_t n_committed, dvn_r1cs** out) try {
    if (!n_committed) return fail(DVN_ERR_INVALID_ARG, "n_committed must be positive");
    return r1cs_synthetic<SuiteBn254>(seed, n_constraints, n_public, out, n_committed, 2);
} DVN_CATCH

// The solver's side of a committed proof, up to the completed witness in pk->d_w (caller holds ctx->mu): upload, the commitment
// hint (two MSMs over the committed wires, in flight together), the keccak challenge into its wire, the constraints that read
// it.  Every rank / device of a shared proof runs this whole: the commitment bases are small next to the key's five sets, and
// replicating them spares an exchange before the challenge (all ranks derive the same challenge from the same commitment).
static int32_t committed_prepare(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                 uint64_t commitment_out[8], uint64_t pok_out[8], uint64_t challenge_out[4]) {
    typedef SuiteBn254 SU;
    typedef FrCfg S;
    typedef dvn_host::HG1 H1;
    constexpr u32 EW = 8;
    hipStream_t st = ctx->stream;
    if (flags & 1) HIP_TRY(hipMemcpyAsync(pk->d_w, witness, pk->m * EW * 4, hipMemcpyDeviceToDevice, st));
    else TRY(dvn_h2d(ctx, pk->d_w, witness, pk->m * EW * 4, st));
    // the public wires the challenge hashes (they are the first n_public wires): from the caller's host copy, or in ONE copy back
    // that runs under the two MSMs below
    std::vector<u32> pubw((size_t)pk->n_public * EW);
    for (u32 wi : pk->public_committed) if (wi >= pk->n_public) return fail(DVN_ERR_INVALID_ARG, "internal: committed public wire %u is not a public wire", wi);
    if (flags & 1) HIP_TRY(hipMemcpyAsync(pubw.data(), pk->d_w, pubw.size() * 4, hipMemcpyDeviceToHost, st));
    else memcpy(pubw.data(), witness, pubw.size() * 4);
    // (1) the commitment hint: two MSMs over the committed wires, in flight together
    const u32* d_committed = pk->d_w + (size_t)pk->n_public * EW;
    MsmLane *lc = nullptr, *lp = nullptr;
    H1::X c, k;
    TRY(SU::msm_submit_g1(ctx, pk->Basis, 0, pk->n_committed, d_committed, &lc));
    int32_t rc = SU::msm_submit_g1(ctx, pk->BasisSigma, 0, pk->n_committed, d_committed, &lp);
    if (rc != DVN_OK) { H1::X t; (void)SU::msm_collect_g1(ctx, lc, &t); return rc; }
    rc = SU::msm_collect_g1(ctx, lc, &c);
    const int32_t rc2 = SU::msm_collect_g1(ctx, lp, &k);
    if (rc != DVN_OK) return rc;
    TRY(rc2);
    if (flags & 1) HIP_TRY(hipStreamSynchronize(st));        // the public wires have arrived (long since: the MSMs were queued behind the copy)
    uint64_t j[12];
    H1::write_jac(c, j); memcpy(commitment_out, j, 64); if (H1::is_inf(c)) memset(commitment_out, 0, 64);
    H1::write_jac(k, j); memcpy(pok_out, j, 64); if (H1::is_inf(k)) memset(pok_out, 0, 64);
    // challenge = keccak256(C.x || C.y || publicCommitted...) mod r
    std::vector<uint8_t> buf(64 + 32 * pk->public_committed.size());
    fp_ext_to_be32((const u32*)commitment_out, buf.data()); fp_ext_to_be32((const u32*)commitment_out + 8, buf.data() + 32);
    for (size_t i = 0; i < pk->public_committed.size(); ++i) fr_ext_to_be32(pubw.data() + (size_t)pk->public_committed[i] * EW, buf.data() + 64 + 32 * i);
    uint8_t hsh[32];
    keccak256(buf.data(), buf.size(), hsh);
    u32 cwv[EW], cext[EW];
    for (int i = 0; i < 8; ++i) { const uint8_t* p = hsh + 4 * (7 - i); cwv[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }
    while (geq8(cwv, FR_MOD_W)) sub8(cwv, FR_MOD_W);
    fr_to_ext_words<S>(fe_from_canonical<S>(fe_unpack<S>(cwv)), cext);
    if (challenge_out) memcpy(challenge_out, cext, 32);
    // the solver's last step: challenge into its wire, then the constraints that read it
    HIP_TRY(hipMemcpyAsync(pk->d_w + (size_t)pk->commit_wire * EW, cext, 32, hipMemcpyHostToDevice, st));
    k_r1cs_patch_rows<S><<<1, 64, 0, st>>>(pk->d_ptr[0], pk->d_idx[0], pk->d_coef[0], pk->d_ptr[1], pk->d_idx[1], pk->d_coef[1], pk->d_ptr[2], pk->d_idx[2],
                                           pk->d_w, (u32)(pk->n_constraints - pk->tail), (u32)pk->n_constraints);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));                 // cext lives on this stack frame
    return DVN_OK;
}
static int32_t committed_key_check(const dvn_groth16_pk* pk) {
    if (pk->suite != SuiteBn254::SUITE) return fail(DVN_ERR_UNSUPPORTED, "committed proofs exist for BN254 keys only");
    if (!pk->n_committed || pk->imported) return fail(DVN_ERR_INVALID_ARG, "the key has no commitment: use dvn_groth16_prove");
    return DVN_OK;
}

// witness: the solver's assignment BEFORE the commitment is known (commitment wire and the constraints reading it are
// placeholders; dvn_r1cs_witness of a committed instance returns exactly that).  commitment_out / pok_out: G1Affine images
// (gnark Proof.Commitments[0], Proof.CommitmentPok); challenge_out: the commitment wire's value (Montgomery words).
extern "C" int32_t dvn_groth16_prove_committed(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                               const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32],
                                               uint64_t commitment_out[8], uint64_t pok_out[8], uint64_t challenge_out[4]) try {
    typedef SuiteBn254 SU;
    TRY(enter(ctx));
    if (!pk || !witness || !r_mont || !s_mont || !proof_out || !commitment_out || !pok_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    TRY(committed_key_check(pk));
    const HostKey<SU>& hk = *(const HostKey<SU>*)pk->hk;
    std::lock_guard<std::mutex> lk(ctx->mu);
    Blind<SU> bl(hk, r_mont, s_mont);
    TRY(committed_prepare(ctx, pk, witness, flags, commitment_out, pok_out, challenge_out));
    MsmResults<SU> R;
    TRY(groth16_device<SU>(ctx, pk, (const uint64_t*)pk->d_w, 1u | 2u, 0, 1, &R));
    groth16_assemble<SU>(hk, R, bl, proof_out);
    return DVN_OK;
} DVN_CATCH

// The statetransition shape shared between the GPUs of a node (the proof the node serialises on,
// /root/reference/sequencer/statetransition.go:80,248; one commitment: /root/reference/config/statetransition_vkey.sol:97-114).
// Rank `rank` of `world`: the commitment hint, challenge and witness completion whole (replicated), then its entry share of the
// five MSMs.  partial_out as dvn_groth16_prove_shard; every rank returns the same commitment, PoK and challenge;
// dvn_groth16_prove_combine assembles the proof from the gathered partial results.
extern "C" int32_t dvn_groth16_prove_committed_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                                     uint32_t rank, uint32_t world, uint64_t partial_out[72],
                                                     uint64_t commitment_out[8], uint64_t pok_out[8], uint64_t challenge_out[4]) try {
    typedef SuiteBn254 SU;
    typedef SU::H1 H1;
    typedef SU::H2 H2;
    TRY(enter(ctx));
    if (!pk || !witness || !partial_out || !commitment_out || !pok_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (world < 1 || world > 64 || rank >= world) return fail(DVN_ERR_INVALID_ARG, "rank %u outside world %u", rank, world);
    TRY(committed_key_check(pk));
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(committed_prepare(ctx, pk, witness, flags, commitment_out, pok_out, challenge_out));
    MsmResults<SU> R;
    TRY(groth16_device<SU>(ctx, pk, (const uint64_t*)pk->d_w, 1u | 2u, rank, world, &R));
    constexpr size_t J1 = PartialLayout<SU>::J1;
    H1::write_jac(R.mA, partial_out); H1::write_jac(R.mB, partial_out + J1); H1::write_jac(R.mK, partial_out + 2 * J1);
    H1::write_jac(R.mZ, partial_out + 3 * J1); H2::write_jac(R.mB2, partial_out + 4 * J1);
    return DVN_OK;
} DVN_CATCH
// ... and over the devices of a multi-device context from one call: keys[i] = the committed key on dvn_ctx_device(ctx, i).
extern "C" int32_t dvn_groth16_prove_committed_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* witness,
                                                     const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32],
                                                     uint64_t commitment_out[8], uint64_t pok_out[8], uint64_t challenge_out[4]) try {
    if (!commitment_out || !pok_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (keys) for (uint32_t i = 0; i < nkeys; ++i) if (keys[i]) TRY(committed_key_check(keys[i]));
    // per device: prepare on its own copy of the key, then its share; device 0's commitment is the one handed back
    std::vector<uint64_t> cm((size_t)nkeys * 20);
    auto prepare = [&](dvn_ctx* ci, dvn_groth16_pk* pk, uint32_t i) -> int32_t {
        return committed_prepare(ci, pk, witness, 0u, cm.data() + (size_t)i * 20, cm.data() + (size_t)i * 20 + 8, cm.data() + (size_t)i * 20 + 16);
    };
    TRY((groth16_prove_multi<SuiteBn254>(ctx, keys, nkeys, witness, 0u, nullptr, 0, r_mont, s_mont, proof_out, prepare)));
    for (uint32_t i = 1; i < nkeys; ++i)
        if (memcmp(cm.data(), cm.data() + (size_t)i * 20, 20 * 8)) return fail(DVN_ERR_DEVICE, "device %u derived a different commitment than device 0", i);
    memcpy(commitment_out, cm.data(), 64); memcpy(pok_out, cm.data() + 8, 64);
    if (challenge_out) memcpy(challenge_out, cm.data() + 16, 32);
    return DVN_OK;
} DVN_CATCH
// uint64 words of the proof image the prover-package calls write for this key: Ar | Bs | Krs of the key's curve, followed by
// Commitments[0] | CommitmentPok (two G1Affine images) when the key carries a BSB22 commitment
extern "C" int32_t dvn_groth16_proof_words(const dvn_groth16_pk* pk, size_t* words) try {
    if (!pk || !words) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const size_t a1 = pk->g1_bytes / 8, a2 = pk->g2_bytes / 8;
    *words = 2 * a1 + a2 + ((pk->n_committed && !pk->imported) ? 2 * a1 : 0);
    return DVN_OK;
} DVN_CATCH
