// EIP-4844 / EIP-7594 blob KZG on the device -- the step right before every state-transition proof:
//   Blob.ComputeCommitment      /root/reference/types/blobs.go:88-95     (gethkzg.BlobToCommitment)
//   Blob.ComputeProof(z)        /root/reference/types/blobs.go:120-133   (gethkzg.ComputeProof -> proof, claim)
//   Blob.ComputeCellProofs      /root/reference/types/blobs.go:97-104    (gethkzg.ComputeCellProofs, 128 proofs)
//   EvaluateBarycentricNative   /root/reference/crypto/blobs/barycentric.go:24-170
// called from BlobEvalData.Set (/root/reference/crypto/blobs/blob.go:40-90).  Included by dvn_bls12381.hip so the MSM
// kernels of G1_381 are instantiated once.
//
// MI355X-first shape: the setup is decompressed ON the device (one square root per thread) and stays resident; a blob costs
// one 128 KiB upload; the quotient of compute_proof is one single-block kernel (batch inversion per thread, LDS tree sums);
// the 128 cell proofs are ONE batched MSM -- 128 quotient polynomials (synthetic division by X^64 - h_k^64, one thread per
// (cell, residue class)) committed over the same 4032 monomial points, every (cell, window) pair a virtual window of one
// sorted entry list -- instead of FK20's Toeplitz/EC-FFT pipeline: KZG proofs are unique group elements, the bytes are the
// same.  The result points (24 KiB for 128 proofs) are normalised with one shared inversion and ZCash-compressed on the host.
#pragma once
#include "suite_api.h"

namespace kzg {
constexpr u32 N = 4096, LOGN = 12, CELLS = 128, CELL = 64, QN = N - CELL;     // QN = coefficients of a cell quotient
typedef Fr381Cfg S;
typedef Fp381Cfg B;
typedef Fe<S> Fr;
typedef Fe<B> Fq;
typedef G1_381 Cv;
constexpr int SW = S::EW, BW = B::EW;       // 8 / 12 u32 words

enum : u32 { ERR_ENCODING = 1, ERR_NOT_ON_CURVE = 2, ERR_NON_CANONICAL = 4 };

__device__ __forceinline__ u32 bitrev(u32 i, u32 bits) { return __brev(i) >> (32 - bits); }
template <class C> __device__ __forceinline__ bool is_canonical_int(const Fe<C>& x) {      // integer limbs: x < modulus
    const Fe<C> c = fe_canonical<C>(x);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d |= c.v[i] ^ x.v[i];
    return d == 0;
}
template <class C> __device__ __forceinline__ Fe<C> to_integer(const Fe<C>& a) { return fe_canonical<C>(fe_mul<C>(a, fe_small<C>(1))); }   // internal -> canonical integer limbs
template <class C> __device__ __forceinline__ Fe<C> pow_bits(const Fe<C>& a, u32 e, int bits) {
    Fe<C> acc = fe_one<C>();
    for (int i = bits - 1; i >= 0; --i) { acc = fe_sqr<C>(acc); if ((e >> i) & 1) acc = fe_mul<C>(acc, a); }
    return acc;
}
// big-endian bytes -> little-endian u32 words
template <int W> __device__ __forceinline__ void be_to_words(const uint8_t* b, u32 (&w)[W]) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint8_t* p = b + 4 * (W - 1 - i);
        w[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
    }
}
// y > (p - 1)/2  <=>  2y > p (p odd), y a canonical integer
__device__ __forceinline__ bool lexicographically_largest(const Fq& y) {
    i32 d[B::NL];
#pragma unroll
    for (int i = 0; i < B::NL; ++i) d[i] = (i32)(2 * y.v[i]) - (i32)B::p(i);
    fe_carry<B>(d);
    return d[B::NL - 1] >= 0;
}

// ---- setup: 48-byte ZCash-compressed G1 -> internal affine image (optionally at the bit-reversed position) --------------
// No subgroup check (the setup is the trusted ceremony output; the reference loads its proving keys with UnsafeReadFrom too,
// /root/reference/circuits/artifacts.go:391-406).
static __global__ void k_g1_decompress(const uint8_t* __restrict__ in, u32* __restrict__ pts, u32 n, u32 brp_bits, u32* __restrict__ err) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + (size_t)48 * i;
    const u32 dst = brp_bits ? bitrev(i, brp_bits) : i;
    u32* out = pts + (size_t)dst * Cv::AW;
    const uint8_t flags = b[0];
    u32 w[BW];
    be_to_words<BW>(b, w);
    w[BW - 1] &= 0x1fffffffu;
    if (!(flags & 0x80)) { atomicOr(err, (u32)ERR_ENCODING); return; }
    if (flags & 0x40) {
        u32 any = (flags & 0x20) ? 1u : 0u;
#pragma unroll
        for (int k = 0; k < BW; ++k) any |= w[k];
        if (any) atomicOr(err, (u32)ERR_ENCODING);
        for (int k = 0; k < Cv::AW; ++k) out[k] = 0;               // affine infinity = all-zero image
        return;
    }
    const Fq xi = fe_unpack<B>(w);
    if (!is_canonical_int<B>(xi)) { atomicOr(err, (u32)ERR_ENCODING); return; }
    const Fq x = fe_from_canonical<B>(xi);
    const Fq one = fe_one<B>(), two = fe_add<B>(one, one), four = fe_add<B>(two, two);
    const Fq y2 = fe_add<B>(fe_mul<B>(fe_sqr<B>(x), x), four);
    // sqrt: y2^((p+1)/4), p = 3 mod 4; exponent limbs built from the modulus table like fe_inv's
    Fq y = one;
    {
        u32 e[B::NL];                                                // (p + 1) >> 2 as 29-bit limbs
        u32 carry = 1;
#pragma unroll
        for (int k = 0; k < B::NL; ++k) { const u32 t = B::p(k) + carry; e[k] = t & LMASK; carry = t >> LB; }
#pragma unroll
        for (int k = 0; k < B::NL; ++k) e[k] = (e[k] >> 2) | (k + 1 < B::NL ? (e[k + 1 < B::NL ? k + 1 : 0] & 3u) << (LB - 2) : 0u);
        for (int k = B::NL - 1; k >= 0; --k) {
            u32 limb = 0;
#pragma unroll
            for (int j = 0; j < B::NL; ++j) limb = (j == k) ? e[j] : limb;
            for (int bt = LB - 1; bt >= 0; --bt) { y = fe_sqr<B>(y); if ((limb >> bt) & 1) y = fe_mul<B>(y, y2); }
        }
    }
    if (!fe_eq<B>(fe_sqr<B>(y), y2)) { atomicOr(err, (u32)ERR_NOT_ON_CURVE); return; }
    if (lexicographically_largest(to_integer<B>(y)) != (bool)(flags & 0x20)) y = fe_neg<B>(y);
    u32 o[Cv::AW];
    Cv::store_affine_internal(Cv::A{x, y}, o);
    store_words16(out, o, Cv::AW);
}

// ---- blob: 4096 x 32 big-endian bytes -> scalar images (x * 2^256 mod r, what the MSM and NTT entry points consume) ------
static __global__ void k_blob_scalars(const uint8_t* __restrict__ blob, u32* __restrict__ f_ext, u32* __restrict__ err) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    u32 w[SW];
    be_to_words<SW>(blob + (size_t)32 * i, w);
    const Fr xi = fe_unpack<S>(w);
    if (!is_canonical_int<S>(xi) || (w[SW - 1] >> 31)) atomicOr(err, (u32)ERR_NON_CANONICAL);     // EIP-4844 bytes_to_bls_field
    u32 o[SW];
    fe_to_ext_mont<S>(fe_from_canonical<S>(xi), o);
    store_words16(f_ext + (size_t)i * SW, o, SW);
}

// LDS tree sum of one field element per thread (blockDim = 1024); every thread gets the total
__device__ __forceinline__ Fr block_sum_fr(Fr v, u32* sh) {
    const u32 t = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < S::NL; ++k) sh[k * 1024 + t] = v.v[k];
    __syncthreads();
    for (u32 s = 512; s >= 1; s >>= 1) {
        if (t < s) {
            Fr a, c;
#pragma unroll
            for (int k = 0; k < S::NL; ++k) { a.v[k] = sh[k * 1024 + t]; c.v[k] = sh[k * 1024 + t + s]; }
            a = fe_add<S>(a, c);
#pragma unroll
            for (int k = 0; k < S::NL; ++k) sh[k * 1024 + t] = a.v[k];
        }
        __syncthreads();
    }
    Fr r;
#pragma unroll
    for (int k = 0; k < S::NL; ++k) r.v[k] = sh[k * 1024];
    return r;
}

// ---- compute_kzg_proof_impl's field part (EIP-4844) in two launches:
//   k_kzg_terms    one thread per element, spread over the chip: w_i = w^bitrev(i), inv_i = 1/(z - w_i) (own Fermat
//                  inversion: no batch trick needed with 4096 lanes), g_i = f_i w_i inv_i; flags m when z = w_m
//   k_kzg_quotient one block: y = f(z) by the barycentric formula of crypto/blobs/barycentric.go:24-170 (or f_m when
//                  z = w_m, :99-107), q_i = (f_i - y)/(w_i - z), and compute_quotient_eval_within_domain for i = m.
// z_int: canonical little-endian words.  consts: w_4096, w_128, 1/4096 (internal).  out_y: canonical integer words of y.
static __global__ void k_kzg_terms(const u32* __restrict__ f_ext, const u32* __restrict__ z_int, const u32* __restrict__ consts,
                                   u32* __restrict__ inv_out, u32* __restrict__ g_out, int* __restrict__ m_out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    u32 zw[SW], cw[SW], w[SW];
#pragma unroll
    for (int k = 0; k < SW; ++k) { zw[k] = z_int[k]; cw[k] = consts[k]; }
    const Fr z = fe_from_canonical<S>(fe_unpack<S>(zw));
    const Fr om = pow_bits<S>(fe_unpack<S>(cw), bitrev(i, LOGN), LOGN);
    load_words16(f_ext + (size_t)i * SW, w, SW);
    const Fr f = fe_from_ext_mont<S>(w);
    const Fr d = fe_sub<S>(z, om);
    const bool hit = fe_is_zero<S>(d);
    if (hit) *m_out = (int)i;
    const Fr inv = fe_inv<S>(d);                                      // 0 -> 0
    const Fr wi = fe_mul<S>(om, inv);                                 // w_i / (z - w_i)
    u32 o[SW];
    fe_pack<S>(fe_canonical<S>(inv), o); store_words16(inv_out + (size_t)i * SW, o, SW);
    fe_pack<S>(fe_canonical<S>(wi), o); store_words16(g_out + (size_t)i * SW, o, SW);
}
static __global__ __launch_bounds__(1024) void k_kzg_quotient(const u32* __restrict__ f_ext, const u32* __restrict__ z_int, const u32* __restrict__ consts,
                                                              const u32* __restrict__ inv_in, const u32* __restrict__ g_in, const int* __restrict__ m_in,
                                                              u32* __restrict__ q_ext, u32* __restrict__ out_y) {
    __shared__ u32 sh[S::NL * 1024];
    const u32 t = threadIdx.x;
    const int m = *m_in;
    u32 zw[SW], nw[SW];
#pragma unroll
    for (int k = 0; k < SW; ++k) { zw[k] = z_int[k]; nw[k] = consts[2 * SW + k]; }
    const Fr z = fe_from_canonical<S>(fe_unpack<S>(zw));
    Fr f[4], inv[4], wi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const u32 i = t + 1024 * e;
        u32 w[SW];
        load_words16(f_ext + (size_t)i * SW, w, SW); f[e] = fe_from_ext_mont<S>(w);
        load_words16(inv_in + (size_t)i * SW, w, SW); inv[e] = fe_unpack<S>(w);
        load_words16(g_in + (size_t)i * SW, w, SW); wi[e] = fe_unpack<S>(w);
    }
    Fr y;
    if (m < 0) {
        Fr acc = fe_zero<S>();
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fe_add<S>(acc, fe_mul<S>(f[e], wi[e]));
        const Fr sum = block_sum_fr(acc, sh);
        Fr zp = z;                                                    // (z^4096 - 1) / 4096
        for (u32 k = 0; k < LOGN; ++k) zp = fe_sqr<S>(zp);
        y = fe_mul<S>(sum, fe_mul<S>(fe_sub<S>(zp, fe_one<S>()), fe_unpack<S>(nw)));
    } else {
        Fr mine = fe_zero<S>();
#pragma unroll
        for (int e = 0; e < 4; ++e) if ((int)(t + 1024 * e) == m) mine = f[e];
        y = block_sum_fr(mine, sh);                                   // exactly one non-zero term: f_m
    }
    Fr q[4], acc2 = fe_zero<S>();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const Fr df = fe_sub<S>(f[e], y);
        q[e] = fe_neg<S>(fe_mul<S>(df, inv[e]));                      // (f_i - y) / (w_i - z); inv_m = 0
        acc2 = fe_add<S>(acc2, fe_mul<S>(df, wi[e]));                 // (f_i - y) w_i / (z - w_i); the i = m term is 0
    }
    if (m >= 0) {                                                     // q_m = sum_{i != m} (f_i - y) w_i / (z (z - w_i))
        const Fr u = block_sum_fr(acc2, sh);
        const Fr qm = fe_mul<S>(u, fe_inv<S>(z));
#pragma unroll
        for (int e = 0; e < 4; ++e) if ((int)(t + 1024 * e) == m) q[e] = qm;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u32 o[SW];
        fe_to_ext_mont<S>(q[e], o);
        store_words16(q_ext + (size_t)(t + 1024 * e) * SW, o, SW);
    }
    if (t == 0) {
        u32 o[SW];
        fe_pack<S>(to_integer<S>(y), o);
#pragma unroll
        for (int k = 0; k < SW; ++k) out_y[k] = o[k];
    }
}

// ---- cell quotients: q_k(X) = (f(X) - I_k(X)) / (X^64 - c_k), c_k = h_k^64 = w_128^bitrev7(k) (EIP-7594 coset_for_cell);
// thread (k, r) walks the residue class j = r mod 64 from the top: q_k[j] = f[j + 64] + c_k q_k[j + 64].
// coef_ext: the blob's monomial coefficients (scalar images); consts + SW: w_128 (internal); q_ext: CELLS x QN scalar images.
static __global__ void k_cell_quotients(const u32* __restrict__ coef_ext, const u32* __restrict__ consts, u32* __restrict__ q_ext) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= CELLS * CELL) return;
    const u32 k = g / CELL, r = g % CELL;
    u32 cw[SW];
#pragma unroll
    for (int i = 0; i < SW; ++i) cw[i] = consts[SW + i];
    const Fr c = pow_bits<S>(fe_unpack<S>(cw), bitrev(k, 7), 7);
    Fr acc = fe_zero<S>();
    for (int m = (int)(QN / CELL) - 1; m >= 0; --m) {
        const u32 j = r + CELL * (u32)m;
        u32 w[SW];
        load_words16(coef_ext + (size_t)(j + CELL) * SW, w, SW);
        acc = fe_add<S>(fe_from_ext_mont<S>(w), fe_mul<S>(c, acc));
        u32 o[SW];
        fe_to_ext_mont<S>(acc, o);
        store_words16(q_ext + ((size_t)k * QN + j) * SW, o, SW);
    }
}

}  // namespace kzg

struct dvn_kzg_setup {
    dvn_points* lagrange_brp = nullptr;      // L_bitrev(i): basis point of the blob's i-th element
    dvn_points* monomial = nullptr;          // [tau^j]G1 (cell proofs), may be NULL
    dvn_domain* dom = nullptr;               // BLS12-381 Fr domain of 4096
    uint8_t* d_bytes = nullptr;              // staging: blob (128 KiB) / compressed points in (192 KiB) / out (6 KiB)
    u32 *d_f = nullptr, *d_q = nullptr, *d_cellq = nullptr, *d_res = nullptr, *d_small = nullptr;   // d_small: z, y, consts, err, m
    u32 *d_inv = nullptr, *d_g = nullptr;                                                          // 1/(z - w_i), w_i/(z - w_i)
    // expanded tables (row w = 2^(c w) * P): all windows of a scalar vector share one bucket set, no window combine
    u32 *tab_lagrange = nullptr, *tab_monomial = nullptr;
    u32 c_single = 0, rows_single = 0, c_cells = 0, rows_cells = 0;
};

static int32_t kzg_points_from_compressed(dvn_ctx* ctx, dvn_kzg_setup* s, const uint8_t* bytes48, u32 brp_bits, dvn_points** out) {
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = 1; p->n = kzg::N; p->suite = SuiteOf<G1_381>::value; p->expand_c = 0; p->nexp = 1;
    u32* err = s->d_small + 64;
    hipError_t e = hipMalloc((void**)&p->d, (size_t)kzg::N * G1_381::AW * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_bytes, bytes48, (size_t)kzg::N * 48, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(err, 0, 4, ctx->stream);
    u32 herr = 0;
    if (e == hipSuccess) {
        kzg::k_g1_decompress<<<kzg::N / 64, 64, 0, ctx->stream>>>(s->d_bytes, p->d, kzg::N, brp_bits, err);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || herr) {
        if (p->d) (void)hipFree(p->d);
        delete p; (void)hipGetLastError();
        if (e != hipSuccess) return fail(DVN_ERR_DEVICE, "setup decompression failed: %s", hipGetErrorString(e));
        return fail(DVN_ERR_INVALID_ARG, "trusted setup: %s", (herr & kzg::ERR_NOT_ON_CURVE) ? "point is not on the curve" : "bad compressed G1 encoding");
    }
    *out = p;
    return DVN_OK;
}

static void kzg_setup_destroy(dvn_kzg_setup* s) {            // no locking, no ctx: direct frees (caller synchronised)
    if (!s) return;
    for (dvn_points* p : {s->lagrange_brp, s->monomial}) if (p) { if (p->d) (void)hipFree(p->d); delete p; }
    ntt_domain_release(s->dom);
    void* bufs[] = {s->d_bytes, s->d_f, s->d_q, s->d_cellq, s->d_res, s->d_small, s->d_inv, s->d_g, s->tab_lagrange, s->tab_monomial};
    for (void* b : bufs) if (b) (void)hipFree(b);
    delete s;
}
extern "C" int32_t dvn_kzg_setup_free(dvn_ctx* ctx, dvn_kzg_setup* s) try {
    TRY(enter(ctx));
    if (!s) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipDeviceSynchronize();
    kzg_setup_destroy(s);
    return DVN_OK;
} DVN_CATCH

extern "C" int32_t dvn_kzg_setup_load(dvn_ctx* ctx, const uint8_t* g1_lagrange, const uint8_t* g1_monomial, dvn_kzg_setup** out) try {
    TRY(enter(ctx));
    if (!g1_lagrange || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    dvn_kzg_setup* s = new (std::nothrow) dvn_kzg_setup();
    if (!s) return fail(DVN_ERR_OOM, "host allocation failed");
    int32_t rc = ntt_domain_create<Fr381Cfg>(ctx, kzg::LOGN, &s->dom);           // takes ctx->mu itself
    if (rc != DVN_OK) { kzg_setup_destroy(s); return rc; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto alloc = [&](void** p, size_t bytes) -> int32_t { HIP_TRY(hipMalloc(p, bytes)); return DVN_OK; };
    if ((rc = alloc((void**)&s->d_bytes, (size_t)kzg::N * 48)) || (rc = alloc((void**)&s->d_f, (size_t)kzg::N * kzg::SW * 4)) ||
        (rc = alloc((void**)&s->d_q, (size_t)kzg::N * kzg::SW * 4)) || (rc = alloc((void**)&s->d_res, (size_t)kzg::CELLS * G1_381::XW * 4)) ||
        (rc = alloc((void**)&s->d_small, 512)) || (rc = alloc((void**)&s->d_inv, (size_t)kzg::N * kzg::SW * 4)) || (rc = alloc((void**)&s->d_g, (size_t)kzg::N * kzg::SW * 4)) ||
        (g1_monomial && (rc = alloc((void**)&s->d_cellq, (size_t)kzg::CELLS * kzg::QN * kzg::SW * 4)))) { kzg_setup_destroy(s); return rc; }
    // constants: w_4096, w_128, 1/4096 in internal form at d_small + 16 / + 24 / + 32 (u32 words); err at + 64, m at + 65
    u32 h[3 * kzg::SW];
    fr_store_internal<Fr381Cfg>(fr_root_of_unity<Fr381Cfg>(12), h);
    fr_store_internal<Fr381Cfg>(fr_root_of_unity<Fr381Cfg>(7), h + kzg::SW);
    fr_store_internal<Fr381Cfg>(fe_inv<Fr381Cfg>(fr_from_u64<Fr381Cfg>(kzg::N)), h + 2 * kzg::SW);
    hipError_t e = hipMemcpy(s->d_small + 16, h, sizeof(h), hipMemcpyHostToDevice);
    if (e != hipSuccess) { kzg_setup_destroy(s); (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "constant upload failed: %s", hipGetErrorString(e)); }
    if ((rc = kzg_points_from_compressed(ctx, s, g1_lagrange, kzg::LOGN, &s->lagrange_brp)) ||
        (g1_monomial && (rc = kzg_points_from_compressed(ctx, s, g1_monomial, 0, &s->monomial)))) { kzg_setup_destroy(s); return rc; }
    // expanded tables, once per setup: window sizes from the sweep in tools/gpu_kzg_perf.py (DVN_KZG_C / DVN_KZG_CELL_C override)
    auto env_c = [](const char* name, u32 dflt) { const char* v = getenv(name); const int x = v ? atoi(v) : 0; return (u32)((x >= 4 && x <= 16) ? x : dflt); };
    s->c_single = env_c("DVN_KZG_C", 13); s->c_cells = env_c("DVN_KZG_CELL_C", 13);
    if ((rc = msm_row_table_create<G1_381>(ctx, s->lagrange_brp, kzg::N, s->c_single, &s->rows_single, &s->tab_lagrange)) ||
        (g1_monomial && (rc = msm_row_table_create<G1_381>(ctx, s->monomial, kzg::QN, s->c_cells, &s->rows_cells, &s->tab_monomial)))) { kzg_setup_destroy(s); return rc; }
    *out = s;
    return DVN_OK;
} DVN_CATCH

// blob bytes -> device scalar images in s->d_f (caller holds ctx->mu)
static int32_t kzg_load_blob(dvn_ctx* ctx, const dvn_kzg_setup* s, const uint8_t* blob) {
    u32* err = s->d_small + 64;
    u32 herr = 0;
    HIP_TRY(hipMemcpyAsync(s->d_bytes, blob, (size_t)kzg::N * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(err, 0, 4, ctx->stream));
    kzg::k_blob_scalars<<<kzg::N / 256, 256, 0, ctx->stream>>>(s->d_bytes, s->d_f, err);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (herr) return fail(DVN_ERR_INVALID_ARG, "blob element is not a canonical field element");
    return DVN_OK;
}
// ---- host finishing of the result points (host_fast.h, 6 x 64-bit limbs): affine normalisation with ONE shared inversion
// (Montgomery's trick over ZZ_i * ZZZ_i) and ZCash compression.  A 381-bit Fermat inversion is ~25 us here against 0.6 ms for a GPU
// lane, and 128 results are 24 KiB: cheaper to bring back than to invert on the device.  xyzz: n internal images (4 x 12 u32 words).
static void kzg_compress_xyzz_host(const u32* xyzz, u32 n, uint8_t* out48) {
    typedef dvn_host::OpsFqN<Fp381Cfg, 6> O;
    typedef dvn_host::Group<O> HG;
    std::vector<HG::X> pt(n);
    std::vector<O::F> pre(n);                                  // prefix products of d_i = ZZ_i * ZZZ_i over the finite points
    O::F run = O::one();
    for (u32 i = 0; i < n; ++i) {
        pt[i] = HG::load_device(xyzz + (size_t)i * G1_381::XW);
        pre[i] = run;
        if (!HG::is_inf(pt[i])) run = O::mul(run, O::mul(pt[i].zz, pt[i].zzz));
    }
    O::F inv_run = O::inv(run);
    O::F one_int = O::zero(); one_int.l[0] = 1;
    for (int i = (int)n - 1; i >= 0; --i) {
        uint8_t* out = out48 + (size_t)48 * i;
        if (HG::is_inf(pt[i])) { memset(out, 0, 48); out[0] = 0xC0; continue; }
        const O::F dinv = O::mul(inv_run, pre[i]);                                    // 1 / (ZZ_i ZZZ_i)
        inv_run = O::mul(inv_run, O::mul(pt[i].zz, pt[i].zzz));
        const O::F x = O::mul(O::mul(pt[i].x, O::mul(dinv, pt[i].zzz)), one_int);     // X / ZZ, canonical integer
        const O::F y = O::mul(O::mul(pt[i].y, O::mul(dinv, pt[i].zz)), one_int);      // Y / ZZZ
        for (int j = 0; j < 6; ++j) for (int k = 0; k < 8; ++k) out[8 * (5 - j) + (7 - k)] = (uint8_t)(x.l[j] >> (8 * k));
        const O::F y2 = O::add(y, y);                                                   // 2y mod p is odd iff 2y > p (p odd)
        const bool largest = !O::is_zero(y) && (y2.l[0] & 1);
        out[0] |= 0x80 | (largest ? 0x20 : 0x00);
    }
}

// `batch` MSMs of n scalars each over pts[0, n) (expanded table `tab`, window c) -> `batch` compressed points in out48
// (caller holds ctx->mu)
static int32_t kzg_commit(dvn_ctx* ctx, const dvn_kzg_setup* s, const dvn_points* pts, const u32* tab, u32 c, size_t n, const u32* d_scalars, u32 batch, uint8_t* out48) {
    MsmLane* L = nullptr;
    TRY(msm_batched_submit<G1_381>(ctx, pts, 0, n, d_scalars, batch, c, s->d_res, &L, tab));
    std::vector<u32> h((size_t)batch * G1_381::XW);
    hipError_t e1 = hipMemcpyAsync(h.data(), s->d_res, h.size() * 4, hipMemcpyDeviceToHost, L->stream);
    int32_t rc = lane_wait(ctx, *L);
    if (e1 != hipSuccess) { (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "result copy failed: %s", hipGetErrorString(e1)); }
    TRY(rc);
    kzg_compress_xyzz_host(h.data(), batch, out48);
    return rc;
}

extern "C" int32_t dvn_kzg_blob_to_commitment(dvn_ctx* ctx, const dvn_kzg_setup* s, const uint8_t* blob, uint8_t out[48]) try {
    TRY(enter(ctx));
    if (!s || !blob || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(kzg_load_blob(ctx, s, blob));
    return kzg_commit(ctx, s, s->lagrange_brp, s->tab_lagrange, s->c_single, kzg::N, s->d_f, 1, out);
} DVN_CATCH

static int32_t kzg_scalar_from_be(const uint8_t z_be[32], u32 zw[8]) {
    static const u32 R381[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    for (int i = 0; i < 8; ++i) { const uint8_t* p = z_be + 4 * (7 - i); zw[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }
    if (geq8(zw, R381)) return fail(DVN_ERR_INVALID_ARG, "evaluation point is not a canonical field element");
    return DVN_OK;
}
// quotient evaluations into s->d_q, y (big-endian) into y_out (caller holds ctx->mu; blob already loaded)
static int32_t kzg_quotient(dvn_ctx* ctx, const dvn_kzg_setup* s, const u32 zw[8], uint8_t y_out[32]) {
    u32 yw[8];
    HIP_TRY(hipMemcpyAsync(s->d_small, zw, 32, hipMemcpyHostToDevice, ctx->stream));
    int* d_m = (int*)(s->d_small + 65);
    HIP_TRY(hipMemsetAsync(d_m, 0xff, 4, ctx->stream));                 // m = -1: z outside the domain
    kzg::k_kzg_terms<<<kzg::N / 64, 64, 0, ctx->stream>>>(s->d_f, s->d_small, s->d_small + 16, s->d_inv, s->d_g, d_m);
    kzg::k_kzg_quotient<<<1, 1024, 0, ctx->stream>>>(s->d_f, s->d_small, s->d_small + 16, s->d_inv, s->d_g, d_m, s->d_q, s->d_small + 8);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(yw, s->d_small + 8, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i) { uint8_t* p = y_out + 4 * (7 - i); p[0] = (uint8_t)(yw[i] >> 24); p[1] = (uint8_t)(yw[i] >> 16); p[2] = (uint8_t)(yw[i] >> 8); p[3] = (uint8_t)yw[i]; }
    return DVN_OK;
}

extern "C" int32_t dvn_kzg_evaluate_barycentric(dvn_ctx* ctx, const dvn_kzg_setup* s, const uint8_t* blob, const uint8_t z[32], uint8_t y[32]) try {
    TRY(enter(ctx));
    if (!s || !blob || !z || !y) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    u32 zw[8];
    TRY(kzg_scalar_from_be(z, zw));
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(kzg_load_blob(ctx, s, blob));
    return kzg_quotient(ctx, s, zw, y);
} DVN_CATCH

extern "C" int32_t dvn_kzg_compute_proof(dvn_ctx* ctx, const dvn_kzg_setup* s, const uint8_t* blob, const uint8_t z[32], uint8_t proof[48], uint8_t y[32]) try {
    TRY(enter(ctx));
    if (!s || !blob || !z || !proof || !y) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    u32 zw[8];
    TRY(kzg_scalar_from_be(z, zw));
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(kzg_load_blob(ctx, s, blob));
    TRY(kzg_quotient(ctx, s, zw, y));
    return kzg_commit(ctx, s, s->lagrange_brp, s->tab_lagrange, s->c_single, kzg::N, s->d_q, 1, proof);
} DVN_CATCH

// ---- SHA-256 (host): the Fiat-Shamir challenge of compute_blob_kzg_proof --------------------------------------------------
static void kzg_sha256(const uint8_t* data, size_t len, uint8_t out[32]) {
    static const u32 K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    u32 h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::vector<uint8_t> p(data, data + len);
    p.push_back(0x80);
    while (p.size() % 64 != 56) p.push_back(0);
    const u64 bits = (u64)len * 8;
    for (int i = 7; i >= 0; --i) p.push_back((uint8_t)(bits >> (8 * i)));
    auto ror = [](u32 v, int n) { return (v >> n) | (v << (32 - n)); };
    for (size_t off = 0; off < p.size(); off += 64) {
        u32 w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((u32)p[off + 4 * i] << 24) | ((u32)p[off + 4 * i + 1] << 16) | ((u32)p[off + 4 * i + 2] << 8) | p[off + 4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const u32 s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const u32 t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const u32 t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}
extern "C" int32_t dvn_sha256(const uint8_t* data, size_t len, uint8_t out[32]) try {
    if ((!data && len) || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    kzg_sha256(data, len, out);
    return DVN_OK;
} DVN_CATCH

// Blob.ComputeBlobProof (/root/reference/types/blobs.go:108-117, gethkzg.ComputeBlobProof): EIP-4844 compute_blob_kzg_proof --
// the opening proof at the Fiat-Shamir point z = sha256("FSBLOBVERIFY_V1_" || uint128_be(4096) || blob || commitment) mod r
// (compute_challenge).  Like the reference's method it does not check that the commitment belongs to the blob.
extern "C" int32_t dvn_kzg_compute_blob_proof(dvn_ctx* ctx, const dvn_kzg_setup* s, const uint8_t* blob, const uint8_t commitment[48], uint8_t proof[48]) try {
    TRY(enter(ctx));
    if (!s || !blob || !commitment || !proof) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    std::vector<uint8_t> buf(32 + (size_t)kzg::N * 32 + 48, 0);
    memcpy(buf.data(), "FSBLOBVERIFY_V1_", 16);
    buf[16 + 14] = (uint8_t)(kzg::N >> 8); buf[16 + 15] = (uint8_t)kzg::N;          // 16-byte big-endian FIELD_ELEMENTS_PER_BLOB
    memcpy(buf.data() + 32, blob, (size_t)kzg::N * 32);
    memcpy(buf.data() + 32 + (size_t)kzg::N * 32, commitment, 48);
    uint8_t hsh[32], y[32];
    kzg_sha256(buf.data(), buf.size(), hsh);
    static const u32 R381[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    u32 zw[8];
    for (int i = 0; i < 8; ++i) { const uint8_t* p = hsh + 4 * (7 - i); zw[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }
    while (geq8(zw, R381)) sub8(zw, R381);
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(kzg_load_blob(ctx, s, blob));
    TRY(kzg_quotient(ctx, s, zw, y));
    return kzg_commit(ctx, s, s->lagrange_brp, s->tab_lagrange, s->c_single, kzg::N, s->d_q, 1, proof);
} DVN_CATCH

extern "C" int32_t dvn_kzg_compute_cell_proofs(dvn_ctx* ctx, const dvn_kzg_setup* s, const uint8_t* blob, uint8_t* proofs /* 128 x 48 */) try {
    TRY(enter(ctx));
    if (!s || !blob || !proofs) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (!s->monomial) return fail(DVN_ERR_INVALID_ARG, "setup was loaded without the monomial G1 basis: no cell proofs");
    std::lock_guard<std::mutex> lk(ctx->mu);
    TRY(kzg_load_blob(ctx, s, blob));
    // monomial coefficients: the blob lists evaluations in bit-reversed order = the input order of FFTInverse(DIT)
    TRY(ntt_device<Fr381Cfg>(ctx, s->dom, s->d_f, 1, DVN_DIT, 0));
    kzg::k_cell_quotients<<<kzg::CELLS * kzg::CELL / 64, 64, 0, ctx->stream>>>(s->d_f, s->d_small + 16, s->d_cellq);
    HIP_TRY(hipGetLastError());
    return kzg_commit(ctx, s, s->monomial, s->tab_monomial, s->c_cells, kzg::QN, s->d_cellq, kzg::CELLS, proofs);
} DVN_CATCH
