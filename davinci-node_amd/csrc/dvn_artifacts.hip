// gnark proving-key artifacts on BN254 (the statetransition and results keys) + the context's sha256-keyed key cache.
// The reader itself is curve-generic: artifacts_impl.h (layout, point codecs, device decompression, cache protocol); the
// BLS12-377 and BW6-761 instantiations live next to their suites in dvn_bls12377.hip / dvn_bw6761.hip.
#include "artifacts_impl.h"
#include "suite_bn254.h"

namespace dvn_art {
template <> struct KeyTraits<SuiteBn254> {
    static constexpr const char* NAME = "BN254";
    static constexpr u32 COSET_GEN = 5;                                   // fr.MultiplicativeGen of BN254
    static CurveAux1<FpCfg> aux_g1() { return make_aux1<FpCfg>(fe_from_canonical<FpCfg>(fe_small<FpCfg>(3)), 0); }   // y^2 = x^3 + 3
    // twist constant 3 / (9 + u) of BN254 G2 (y^2 = x^3 + b'), canonical words
    static CurveAux2<FpCfg> aux_g2() {
        const u32 a0[8] = {0x24a138e5u, 0x3267e6dcu, 0x59dbefa3u, 0xb5b4c5e5u, 0x1be06ac3u, 0x81be1899u, 0xceb8aaaeu, 0x2b149d40u};
        const u32 a1[8] = {0x85c315d2u, 0xe4a2bd06u, 0xe52d1852u, 0xa74fa084u, 0xeed8fdf4u, 0xcd2cafadu, 0x3af0fed4u, 0x009713b0u};
        return make_aux2<FpCfg>(Fp2{fe_from_canonical<FpCfg>(fe_unpack<FpCfg>(a0)), fe_from_canonical<FpCfg>(fe_unpack<FpCfg>(a1))}, 0);
    }
};
}  // namespace dvn_art

int32_t dvn_internal_key_cache_clear(dvn_ctx* ctx) {
    std::vector<std::pair<std::string, dvn_groth16_pk*>> keys;
    { std::lock_guard<std::mutex> lk(ctx->mu); keys.swap(ctx->key_cache); }
    for (auto& kv : keys) { kv.second->cached = false; (void)dvn_groth16_pk_free(ctx, kv.second); }
    return DVN_OK;
}
// Drop one key from the cache and release it (a node that rotates circuit artifacts would otherwise keep every key it ever
// loaded resident until dvn_shutdown).  Handles to it become invalid.  A key that is not cached is simply freed.
extern "C" int32_t dvn_groth16_pk_evict(dvn_ctx* ctx, dvn_groth16_pk* pk) try {
    TRY(enter(ctx));
    if (!pk) return DVN_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (size_t i = 0; i < ctx->key_cache.size(); ++i)
            if (ctx->key_cache[i].second == pk) { ctx->key_cache.erase(ctx->key_cache.begin() + (long)i); break; }
    }
    pk->cached = false;
    return dvn_groth16_pk_free(ctx, pk);
} DVN_CATCH

// Host only: what the file holds, without touching a device (tests on CPU; also a cheap sanity check before a load).
// info: log_n, raw (0 / 1), n_wires, |A|, |B|, |Z|, |K|, |G2.B|, n_commitments, |Basis[0]|, domain flag byte present (0 / 1),
// []bool bitmaps carry a uint32 length prefix (0 / 1)
extern "C" int32_t dvn_gnark_pk_scan(const uint8_t* data, size_t len, uint64_t info[12]) try { return dvn_art::pk_scan_api<SuiteBn254>(data, len, info); } DVN_CATCH
extern "C" int32_t dvn_groth16_pk_load_bytes(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                             uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    return dvn_art::pk_load_bytes_api<SuiteBn254>(ctx, data, len, n_public, committed_wires, n_committed_wires, out);
} DVN_CATCH
extern "C" int32_t dvn_groth16_pk_load_file(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                            const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    return dvn_art::pk_load_file_api<SuiteBn254>(ctx, path, sha256_hex, n_public, committed_wires, n_committed_wires, out);
} DVN_CATCH
// the commitment bases of a loaded key (NULL when the key has no commitment): what the solver's BSB22 hint passes to
// dvn_pedersen_commit.  The handles stay owned by the key.
extern "C" int32_t dvn_groth16_pk_commitment_bases(const dvn_groth16_pk* pk, const dvn_points** basis, const dvn_points** basis_exp_sigma) try {
    if (!pk || !basis || !basis_exp_sigma) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    *basis = pk->Basis; *basis_exp_sigma = pk->BasisSigma;
    return DVN_OK;
} DVN_CATCH
