// BLS12-381 instantiation: the fourth curve of the reference's GPU dispatch (/root/reference/prover/prover_gpu.go:45-50)
// and the curve of the EIP-4844 blob commitment that precedes every state-transition proof (a
// 4096-point G1 MSM over the KZG setup, /root/reference/crypto/blobs/blob.go:40-90, /root/reference/types/blobs.go:90-160).
// Same kernels: 14-limb base field (381 bits, external image 6 x u64, R = 2^384), Fe2 with u^2 = -1 for G2, and the
// 255-bit scalar field (two-adicity 32, coset shift 7) for the NTT.
#include "suite_api.h"

static G1_381::A g1_381_generator() {      // the standard BLS12-381 G1 generator, Montgomery image (oracle/curves_py.py BLS12_381.g1_gen)
    static const u64 g[12] = {0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL, 0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL,
                              0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL, 0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL};
    return G1_381::affine_from_ext((const u32*)g);
}
static const u32 R381_W[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};

extern "C" int32_t dvn_bls12381_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) { return points_upload<G1_381>(ctx, xy, n, 1, out); }
extern "C" int32_t dvn_bls12381_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) { return points_upload<G2_381>(ctx, xy, n, 2, out); }
extern "C" int32_t dvn_bls12381_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out) {
    return suite_points_synthetic<G1_381>(ctx, 1, g1_381_generator(), R381_W, seed, n, out);
}
extern "C" int32_t dvn_bls12381_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) {
    return suite_points_download<G1_381, G2_381>(ctx, "BLS12-381", pts, off, n, xy);
}
extern "C" int32_t dvn_bls12381_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[18]) {
    return msm_full<G1_381>(ctx, pts, 1, off, n, scalars, flags, out_jac);
}
extern "C" int32_t dvn_bls12381_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[36]) {
    return msm_full<G2_381>(ctx, pts, 2, off, n, scalars, flags, out_jac);
}
extern "C" int32_t dvn_bls12381_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) { return suite_scalars_synthetic<Fr381Cfg>(ctx, seed, kind, n, d_out); }
extern "C" int32_t dvn_bls12381_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) { return ntt_domain_create<Fr381Cfg>(ctx, log_n, out); }
extern "C" int32_t dvn_bls12381_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset, uint32_t flags) {
    return suite_ntt<Fr381Cfg>(ctx, "BLS12-381", dom, data, inverse, decimation, on_coset, flags);
}
extern "C" int32_t dvn_bls12381_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                                 size_t len, uint64_t* h_out, uint32_t flags) {
    return compute_h_api<Fr381Cfg>(ctx, "BLS12-381", dom, a, b, c, len, h_out, flags);
}
