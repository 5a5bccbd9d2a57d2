// gnark proving-key artifact reader (BN254) + sha256-keyed device-resident key cache.
//
// The node keeps its circuit artifacts as sha256-named files and loads a proving key with
//     pk := newProvingKey(curve); pk.UnsafeReadFrom(file)              /root/reference/circuits/artifacts.go:391-406
// after cmd/circuit-compile wrote it with pk.WriteTo (/root/reference/cmd/circuit-compile/main.go:507-513).  This file reads
// the same bytes straight into a device-resident key: points are decompressed ON THE DEVICE (one square root per thread) into the
// internal affine images the MSM kernels use, expanded tables are built, and the handle is what dvn_groth16_prove_solved takes.
//
// Layout read here (gnark backend/groth16/bn254 ProvingKey.WriteTo / WriteRawTo over gnark-crypto's bn254 Encoder; the module
// is not vendored in the reference tree, so the layout is restated from the published format and pinned only by this repo's
// own writer, oracle/gnark_key_format_py.py -- "parity unpinned" by reference data):
//   fft.Domain       uint64 Cardinality | fr CardinalityInv | fr Generator | fr GeneratorInv | fr FrMultiplicativeGen |
//                    fr FrMultiplicativeGenInv (32-byte big-endian canonical each) | [bool withPrecompute, 1 byte]
//   G1.Alpha, G1.Beta, G1.Delta                       one point each
//   G1.A, G1.B, G1.Z, G1.K                            uint32 big-endian length + points
//   G2.Beta, G2.Delta                                 one point each
//   G2.B                                              uint32 length + points
//   uint64 nbWires | uint64 NbInfinityA | uint64 NbInfinityB | nbWires x bool InfinityA | nbWires x bool InfinityB
//   uint32 nbCommitmentKeys, then per key: Basis (uint32 length + points) | BasisExpSigma (uint32 length + points)
// Point encodings (gnark-crypto ecc/bn254/marshal.go): the top two bits of the first byte are 00 uncompressed (G1: X | Y, 64
// bytes; G2: X.A1 | X.A0 | Y.A1 | Y.A0, 128 bytes; infinity = all zero), 10 compressed with the lexicographically smallest y,
// 11 compressed with the largest y, 01 compressed infinity (G1: 32 bytes = X; G2: 64 bytes = X.A1 | X.A0); coordinates are
// big-endian canonical integers.  UnsafeReadFrom performs no subgroup checks; neither does this reader.
#include "groth16_impl.h"
#include "suite_bn254.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" int32_t dvn_sha256(const uint8_t* data, size_t len, uint8_t out[32]);

namespace {

enum : u32 { ERR_ENCODING = 1, ERR_NOT_ON_CURVE = 2, ERR_NOT_CANONICAL = 4 };

template <int W> DVN_HD void be_words(const uint8_t* b, u32 (&w)[W]) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint8_t* p = b + 4 * (W - 1 - i);
        w[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
    }
}
template <class C> DVN_HD bool canonical_int(const Fe<C>& x) {
    const Fe<C> c = fe_canonical<C>(x);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d |= c.v[i] ^ x.v[i];
    return d == 0;
}
template <class C> DVN_HD Fe<C> to_int(const Fe<C>& a) { return fe_canonical<C>(fe_mul<C>(a, fe_small<C>(1))); }
// a^e for e given as 29-bit limbs
template <class C> DVN_HD Fe<C> pow_limbs(const Fe<C>& a, const u32 (&e)[C::NL]) {
    Fe<C> acc = fe_one<C>();
    for (int k = C::NL - 1; k >= 0; --k) {
        u32 limb = 0;
#pragma unroll
        for (int j = 0; j < C::NL; ++j) limb = (j == k) ? e[j] : limb;
        for (int bt = LB - 1; bt >= 0; --bt) { acc = fe_sqr<C>(acc); if ((limb >> bt) & 1) acc = fe_mul<C>(acc, a); }
    }
    return acc;
}
// candidate square root a^((p + 1) / 4) for p = 3 mod 4 (BN254's base field); the caller checks the square
DVN_HD Fp fp_sqrt_candidate(const Fp& a) {
    typedef FpCfg C;
    u32 e[C::NL];
    u32 carry = 1;
#pragma unroll
    for (int k = 0; k < C::NL; ++k) { const u32 t = C::p(k) + carry; e[k] = t & LMASK; carry = t >> LB; }
#pragma unroll
    for (int k = 0; k < C::NL; ++k) e[k] = (e[k] >> 2) | (k + 1 < C::NL ? (e[k + 1 < C::NL ? k + 1 : 0] & 3u) << (LB - 2) : 0u);
    return pow_limbs<C>(a, e);
}
DVN_HD bool fp_sqrt(const Fp& a, Fp* out) {
    const Fp y = fp_sqrt_candidate(a);
    *out = y;
    return fe_eq<FpCfg>(fe_sqr<FpCfg>(y), a);
}
// y > (p - 1) / 2  <=>  2 y > p, y a canonical integer (fp.Element.LexicographicallyLargest)
DVN_HD bool fp_lex_largest(const Fp& y_int) {
    const Fp t = fe_add_lazy<FpCfg>(y_int, y_int);
    for (int i = FpCfg::NL - 1; i >= 0; --i) {
        if (t.v[i] > FpCfg::p(i)) return true;
        if (t.v[i] < FpCfg::p(i)) return false;
    }
    return false;
}
DVN_HD Fp fp_b() { return fe_from_canonical<FpCfg>(fe_small<FpCfg>(3)); }
// twist constant 3 / (9 + u) of BN254 G2 (y^2 = x^3 + b'), canonical words
DVN_HD Fp2 fp2_b() {
    const u32 a0[8] = {0x24a138e5u, 0x3267e6dcu, 0x59dbefa3u, 0xb5b4c5e5u, 0x1be06ac3u, 0x81be1899u, 0xceb8aaaeu, 0x2b149d40u};
    const u32 a1[8] = {0x85c315d2u, 0xe4a2bd06u, 0xe52d1852u, 0xa74fa084u, 0xeed8fdf4u, 0xcd2cafadu, 0x3af0fed4u, 0x009713b0u};
    return Fp2{fe_from_canonical<FpCfg>(fe_unpack<FpCfg>(a0)), fe_from_canonical<FpCfg>(fe_unpack<FpCfg>(a1))};
}
// square root in Fp2 = Fp[u] / (u^2 + 1) with base-field arithmetic only (the "complex" method): for a = a0 + a1 u,
// s = sqrt(a0^2 + a1^2), x0 = sqrt((a0 +- s) / 2), x1 = a1 / (2 x0).  Returns false when a is not a square.
DVN_HD bool fp2_sqrt(const Fp2& a, Fp2* out) {
    typedef FpCfg C;
    typedef Ops<Fp> O;
    typedef Ops<Fp2> O2;
    const Fp zero = fe_zero<C>();
    if (O::is_zero(a.a1)) {
        Fp r;
        if (fp_sqrt(a.a0, &r)) { *out = Fp2{r, zero}; return true; }
        if (fp_sqrt(O::neg(a.a0), &r)) { *out = Fp2{zero, r}; return true; }     // (r u)^2 = -r^2
        return false;
    }
    Fp s;
    if (!fp_sqrt(O::add(O::sqr(a.a0), O::sqr(a.a1)), &s)) return false;
    const Fp half = O::inv(O::add(O::one(), O::one()));
    Fp x0;
    if (!fp_sqrt(O::mul(O::add(a.a0, s), half), &x0)) {
        if (!fp_sqrt(O::mul(O::sub(a.a0, s), half), &x0)) return false;
    }
    const Fp x1 = O::mul(a.a1, O::inv(O::add(x0, x0)));
    const Fp2 r{x0, x1};
    const Fp2 chk = O2::sqr(r);
    if (!fe_eq<C>(chk.a0, a.a0) || !fe_eq<C>(chk.a1, a.a1)) return false;
    *out = r;
    return true;
}
// E2.LexicographicallyLargest: by A1 unless it is zero, then by A0
DVN_HD bool fp2_lex_largest(const Fp2& y) {
    const Fp i1 = to_int<FpCfg>(y.a1);
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < FpCfg::NL; ++i) z |= i1.v[i];
    if (z) return fp_lex_largest(i1);
    return fp_lex_largest(to_int<FpCfg>(y.a0));
}

// one G1 point at `b` (32 bytes compressed, or 64 raw when the flag bits are 00 -> *consumed) into an internal affine image
DVN_HD u32 decode_g1(const uint8_t* b, bool raw_expected, u32* out_words) {
    const u32 flag = b[0] >> 6;
    u32 w[8];
    be_words<8>(b, w);
    w[7] &= 0x3fffffffu;
    if ((flag == 0) != raw_expected) return ERR_ENCODING;
    if (flag == 1) {                                               // compressed infinity: X must be zero
        u32 any = 0;
        for (int k = 0; k < 8; ++k) any |= w[k];
        for (int k = 0; k < G1::AW; ++k) out_words[k] = 0;
        return any ? ERR_ENCODING : 0;
    }
    const Fp xi = fe_unpack<FpCfg>(w);
    if (!canonical_int<FpCfg>(xi)) return ERR_NOT_CANONICAL;
    if (flag == 0) {
        u32 wy[8];
        be_words<8>(b + 32, wy);
        const Fp yi = fe_unpack<FpCfg>(wy);
        if (!canonical_int<FpCfg>(yi)) return ERR_NOT_CANONICAL;
        u32 any = 0;
        for (int k = 0; k < 8; ++k) any |= w[k] | wy[k];
        if (!any) { for (int k = 0; k < G1::AW; ++k) out_words[k] = 0; return 0; }      // uncompressed infinity = (0, 0)
        const Fp x = fe_from_canonical<FpCfg>(xi), y = fe_from_canonical<FpCfg>(yi);
        if (!fe_eq<FpCfg>(fe_sqr<FpCfg>(y), fe_add<FpCfg>(fe_mul<FpCfg>(fe_sqr<FpCfg>(x), x), fp_b()))) return ERR_NOT_ON_CURVE;
        G1::store_affine_internal(G1::A{x, y}, out_words);
        return 0;
    }
    const Fp x = fe_from_canonical<FpCfg>(xi);
    Fp y;
    if (!fp_sqrt(fe_add<FpCfg>(fe_mul<FpCfg>(fe_sqr<FpCfg>(x), x), fp_b()), &y)) return ERR_NOT_ON_CURVE;
    if (fp_lex_largest(to_int<FpCfg>(y)) != (flag == 3)) y = fe_neg<FpCfg>(y);
    G1::store_affine_internal(G1::A{x, y}, out_words);
    return 0;
}
DVN_HD u32 decode_g2(const uint8_t* b, bool raw_expected, u32* out_words) {
    const u32 flag = b[0] >> 6;
    u32 w1[8], w0[8];
    be_words<8>(b, w1); be_words<8>(b + 32, w0);                   // X.A1 first, then X.A0
    w1[7] &= 0x3fffffffu;
    if ((flag == 0) != raw_expected) return ERR_ENCODING;
    if (flag == 1) {
        u32 any = 0;
        for (int k = 0; k < 8; ++k) any |= w1[k] | w0[k];
        for (int k = 0; k < G2::AW; ++k) out_words[k] = 0;
        return any ? ERR_ENCODING : 0;
    }
    const Fp x1i = fe_unpack<FpCfg>(w1), x0i = fe_unpack<FpCfg>(w0);
    if (!canonical_int<FpCfg>(x1i) || !canonical_int<FpCfg>(x0i)) return ERR_NOT_CANONICAL;
    const Fp2 x{fe_from_canonical<FpCfg>(x0i), fe_from_canonical<FpCfg>(x1i)};
    typedef Ops<Fp2> O2;
    const Fp2 rhs = O2::add(O2::mul(O2::sqr(x), x), fp2_b());
    if (flag == 0) {
        u32 v1[8], v0[8];
        be_words<8>(b + 64, v1); be_words<8>(b + 96, v0);
        const Fp y1i = fe_unpack<FpCfg>(v1), y0i = fe_unpack<FpCfg>(v0);
        if (!canonical_int<FpCfg>(y1i) || !canonical_int<FpCfg>(y0i)) return ERR_NOT_CANONICAL;
        u32 any = 0;
        for (int k = 0; k < 8; ++k) any |= w1[k] | w0[k] | v1[k] | v0[k];
        if (!any) { for (int k = 0; k < G2::AW; ++k) out_words[k] = 0; return 0; }
        const Fp2 y{fe_from_canonical<FpCfg>(y0i), fe_from_canonical<FpCfg>(y1i)};
        const Fp2 yy = O2::sqr(y);
        if (!fe_eq<FpCfg>(yy.a0, rhs.a0) || !fe_eq<FpCfg>(yy.a1, rhs.a1)) return ERR_NOT_ON_CURVE;
        G2::store_affine_internal(G2::A{x, y}, out_words);
        return 0;
    }
    Fp2 y;
    if (!fp2_sqrt(rhs, &y)) return ERR_NOT_ON_CURVE;
    if (fp2_lex_largest(y) != (flag == 3)) y = O2::neg(y);
    G2::store_affine_internal(G2::A{x, y}, out_words);
    return 0;
}

__global__ void k_decode_g1(const uint8_t* __restrict__ in, u32* __restrict__ pts, u32 n, u32 stride, u32 raw, u32* __restrict__ err) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[G1::AW];
    const u32 e = decode_g1(in + (size_t)stride * i, raw != 0, w);
    if (e) { atomicOr(err, e); return; }
    store_words16(pts + (size_t)i * G1::AW, w, G1::AW);
}
__global__ void k_decode_g2(const uint8_t* __restrict__ in, u32* __restrict__ pts, u32 n, u32 stride, u32 raw, u32* __restrict__ err) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[G2::AW];
    const u32 e = decode_g2(in + (size_t)stride * i, raw != 0, w);
    if (e) { atomicOr(err, e); return; }
    store_words16(pts + (size_t)i * G2::AW, w, G2::AW);
}

// ---- byte cursor over the file image ------------------------------------------------------------------------------------
struct Cursor {
    const uint8_t* p; size_t len, off = 0; bool ok = true;
    Cursor(const uint8_t* d, size_t l) : p(d), len(l) {}
    const uint8_t* take(size_t k) { if (!ok || k > len - off) { ok = false; return nullptr; } const uint8_t* r = p + off; off += k; return r; }
    uint64_t u64be() { const uint8_t* b = take(8); uint64_t v = 0; if (b) for (int i = 0; i < 8; ++i) v = (v << 8) | b[i]; return v; }
    uint32_t u32be() { const uint8_t* b = take(4); uint32_t v = 0; if (b) for (int i = 0; i < 4; ++i) v = (v << 8) | b[i]; return v; }
};
struct Section { size_t off = 0; uint32_t count = 0; bool raw = false; };      // a point slice: byte offset of its first point
struct Layout {
    uint32_t log_n = 0; bool precompute_flag = false, raw = false;
    size_t alpha = 0, beta = 0, delta = 0, beta2 = 0, delta2 = 0;             // offsets of the single points
    Section A, B, Z, K, B2;
    uint64_t n_wires = 0, n_inf_a = 0, n_inf_b = 0;
    size_t inf_a = 0, inf_b = 0;
    uint32_t n_commitments = 0;
    std::vector<Section> basis, basis_sigma;
    size_t end = 0;
};
// walk the sections; with_flag: the Domain carries the 1-byte withPrecompute bool
static bool scan(const uint8_t* d, size_t len, bool with_flag, Layout* L, std::string* why) {
    Cursor c(d, len);
    const uint64_t card = c.u64be();
    if (!c.ok || card < 2 || (card & (card - 1)) || card > ((uint64_t)1 << 27)) { *why = "domain cardinality is not a power of two in [2, 2^27]"; return false; }
    L->log_n = 0; while (((uint64_t)1 << L->log_n) < card) ++L->log_n;
    c.take(5 * 32);
    L->precompute_flag = with_flag;
    if (with_flag) { const uint8_t* f = c.take(1); if (!f || *f > 1) { *why = "domain flag byte is not a bool"; return false; } }
    if (!c.ok) { *why = "truncated domain header"; return false; }
    const uint8_t* first = d + c.off;
    if (c.off >= len) { *why = "truncated"; return false; }
    L->raw = (first[0] >> 6) == 0;
    const size_t g1 = L->raw ? 64 : 32, g2 = L->raw ? 128 : 64;
    auto single = [&](size_t* where, size_t sz) { *where = c.off; c.take(sz); };
    auto slice = [&](Section* s, size_t sz) { s->count = c.u32be(); s->off = c.off; s->raw = L->raw; c.take((size_t)s->count * sz); };
    single(&L->alpha, g1); single(&L->beta, g1); single(&L->delta, g1);
    slice(&L->A, g1); slice(&L->B, g1); slice(&L->Z, g1); slice(&L->K, g1);
    single(&L->beta2, g2); single(&L->delta2, g2);
    slice(&L->B2, g2);
    L->n_wires = c.u64be(); L->n_inf_a = c.u64be(); L->n_inf_b = c.u64be();
    if (!c.ok || L->n_wires > ((uint64_t)1 << 31)) { *why = "truncated key or implausible wire count"; return false; }
    L->inf_a = c.off; c.take((size_t)L->n_wires);
    L->inf_b = c.off; c.take((size_t)L->n_wires);
    L->n_commitments = c.u32be();
    if (!c.ok || L->n_commitments > 1024) { *why = "truncated key or implausible commitment count"; return false; }
    L->basis.clear(); L->basis_sigma.clear();
    for (uint32_t i = 0; i < L->n_commitments; ++i) {
        Section b, s;
        slice(&b, g1); slice(&s, g1);
        L->basis.push_back(b); L->basis_sigma.push_back(s);
    }
    if (!c.ok) { *why = "truncated key"; return false; }
    L->end = c.off;
    if (L->end != len) { *why = "trailing bytes after the last commitment key"; return false; }
    // consistency the writer guarantees: A holds the wires not flagged in InfinityA, B and G2.B those not flagged in InfinityB
    if (L->A.count + L->n_inf_a != L->n_wires || L->B.count + L->n_inf_b != L->n_wires || L->B2.count != L->B.count) { *why = "point counts disagree with the infinity bitmaps"; return false; }
    if ((uint64_t)L->Z.count + 1 < card) { *why = "G1.Z holds fewer than n - 1 points"; return false; }
    return true;
}
static bool scan_any(const uint8_t* d, size_t len, Layout* L, std::string* why) {
    std::string w1, w2;
    if (scan(d, len, true, L, &w1)) return true;              // gnark-crypto >= 0.12 writes the withPrecompute bool
    if (scan(d, len, false, L, &w2)) return true;
    *why = w1 + " (with the domain flag byte) / " + w2 + " (without)";
    return false;
}

template <class Cv> static int32_t decode_slice(dvn_ctx* ctx, const uint8_t* d_bytes, const Section& s, size_t point_bytes, int curve, dvn_points** out) {
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = curve; p->n = s.count; p->suite = 0;
    expand_plan(s.count, FrCfg::BITS, (size_t)Cv::AW * 4, &p->expand_c, &p->nexp);
    u32* err = nullptr;
    hipError_t e = hipMalloc((void**)&p->d, s.count ? (size_t)s.count * Cv::AW * 4 * p->nexp : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&err, 4);
    if (e == hipSuccess) e = hipMemsetAsync(err, 0, 4, ctx->stream);
    u32 herr = 0;
    if (e == hipSuccess && s.count) {
        if (curve == 1) k_decode_g1<<<(s.count + 63) / 64, 64, 0, ctx->stream>>>(d_bytes + s.off, p->d, s.count, (u32)point_bytes, s.raw ? 1u : 0u, err);
        else k_decode_g2<<<(s.count + 63) / 64, 64, 0, ctx->stream>>>(d_bytes + s.off, p->d, s.count, (u32)point_bytes, s.raw ? 1u : 0u, err);
        if (p->expand_c) k_points_expand<Cv><<<(unsigned)((s.count + 63) / 64), 64, 0, ctx->stream>>>(p->d, s.count, p->expand_c, p->nexp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (err) (void)hipFree(err);
    if (e != hipSuccess || herr) {
        if (p->d) (void)hipFree(p->d);
        delete p; (void)hipGetLastError();
        if (e != hipSuccess) return fail(DVN_ERR_DEVICE, "point decoding failed: %s", hipGetErrorString(e));
        return fail(DVN_ERR_INVALID_ARG, "proving key: %s", (herr & ERR_NOT_ON_CURVE) ? "a point is not on the curve" : (herr & ERR_NOT_CANONICAL) ? "a coordinate is not below the modulus" : "mixed or invalid point encodings in one array");
    }
    *out = p;
    return DVN_OK;
}

static std::string hex32(const uint8_t h[32]) {
    static const char* d = "0123456789abcdef";
    std::string s(64, '0');
    for (int i = 0; i < 32; ++i) { s[2 * i] = d[h[i] >> 4]; s[2 * i + 1] = d[h[i] & 15]; }
    return s;
}

static int32_t load_image(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed, uint32_t n_committed,
                          dvn_groth16_pk** out) {
    typedef FrCfg S;
    constexpr size_t EB = 32;
    Layout L; std::string why;
    if (!scan_any(data, len, &L, &why)) return fail(DVN_ERR_INVALID_ARG, "not a gnark BN254 proving key: %s", why.c_str());
    if (L.n_commitments > 1) return fail(DVN_ERR_UNSUPPORTED, "proving key has %u commitment keys; one is supported (the statetransition circuit has exactly one)", L.n_commitments);
    const size_t m = (size_t)L.n_wires, n = (size_t)1 << L.log_n;
    if (n_public > m) return fail(DVN_ERR_INVALID_ARG, "n_public exceeds the wire count");
    // wire lists: A / B keep the wires not flagged at infinity; K the private wires that are not committed
    std::vector<u32> idxA, idxB, idxK;
    idxA.reserve(L.A.count); idxB.reserve(L.B.count);
    for (size_t i = 0; i < m; ++i) {
        const uint8_t fa = data[L.inf_a + i], fb = data[L.inf_b + i];
        if (fa > 1 || fb > 1) return fail(DVN_ERR_INVALID_ARG, "infinity bitmap byte %zu is not a bool", i);
        if (!fa) idxA.push_back((u32)i);
        if (!fb) idxB.push_back((u32)i);
    }
    if (idxA.size() != L.A.count || idxB.size() != L.B.count) return fail(DVN_ERR_INVALID_ARG, "infinity bitmaps disagree with the point counts");
    std::vector<uint8_t> is_committed(m, 0);
    for (uint32_t i = 0; i < n_committed; ++i) {
        if (!committed || committed[i] >= m || committed[i] < n_public) return fail(DVN_ERR_INVALID_ARG, "committed wire %u is not a private wire", committed ? committed[i] : 0);
        is_committed[committed[i]] = 1;
    }
    for (size_t i = n_public; i < m; ++i) if (!is_committed[i]) idxK.push_back((u32)i);
    if (idxK.size() != L.K.count) return fail(DVN_ERR_INVALID_ARG, "G1.K holds %u points but %zu private non-committed wires follow from n_public = %u and %u committed wires", L.K.count, idxK.size(), n_public, n_committed);
    if (L.n_commitments == 1 && L.basis[0].count != L.basis_sigma[0].count) return fail(DVN_ERR_INVALID_ARG, "commitment key: Basis and BasisExpSigma differ in length");

    dvn_groth16_pk* pk = new (std::nothrow) dvn_groth16_pk();
    if (!pk) return fail(DVN_ERR_OOM, "host allocation failed");
    pk->suite = 0; pk->imported = true; pk->device = ctx->device;
    pk->scalar_bytes = 32; pk->g1_bytes = G1::AW * 4; pk->g2_bytes = G2::AW * 4;
    pk->log_n = L.log_n; pk->n = n; pk->m = m; pk->n_public = n_public; pk->n_constraints = n;
    pk->nA = L.A.count; pk->nB = L.B.count; pk->nK = L.K.count; pk->k_first = m - L.K.count;
    uint8_t* d_bytes = nullptr;
    auto body = [&]() -> int32_t {
        TRY(dvn_ntt_domain(ctx, L.log_n, &pk->dom));
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIP_TRY(hipMalloc((void**)&d_bytes, len));
        HIP_TRY(hipMemcpy(d_bytes, data, len, hipMemcpyHostToDevice));
        const size_t g1 = L.raw ? 64 : 32, g2 = L.raw ? 128 : 64;
        TRY((decode_slice<G1>(ctx, d_bytes, L.A, g1, 1, &pk->A)));
        TRY((decode_slice<G1>(ctx, d_bytes, L.B, g1, 1, &pk->B)));
        TRY((decode_slice<G1>(ctx, d_bytes, L.Z, g1, 1, &pk->Z)));
        TRY((decode_slice<G1>(ctx, d_bytes, L.K, g1, 1, &pk->K)));
        TRY((decode_slice<G2>(ctx, d_bytes, L.B2, g2, 2, &pk->B2)));
        if (L.n_commitments == 1) {
            TRY((decode_slice<G1>(ctx, d_bytes, L.basis[0], g1, 1, &pk->Basis)));
            TRY((decode_slice<G1>(ctx, d_bytes, L.basis_sigma[0], g1, 1, &pk->BasisSigma)));
            pk->n_committed = L.basis[0].count;
        }
        auto up = [&](u32** dst, const std::vector<u32>& v) -> int32_t {
            HIP_TRY(hipMalloc((void**)dst, (v.empty() ? 1 : v.size()) * 4));
            if (!v.empty()) HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice));
            return DVN_OK;
        };
        TRY(up(&pk->d_idxA, idxA)); TRY(up(&pk->d_idxB, idxB)); TRY(up(&pk->d_idxK, idxK));
        HIP_TRY(hipMalloc((void**)&pk->d_w, m * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_abc, 3 * n * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wA, (pk->nA ? pk->nA : 1) * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wB, (pk->nB ? pk->nB : 1) * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wK, (pk->nK ? pk->nK : 1) * EB));
        // the five key constants: decoded on the host with the same routine the kernels run
        typedef dvn_host::HG1 H1;
        typedef dvn_host::HG2 H2;
        HostKey<SuiteBn254>* hk = new HostKey<SuiteBn254>();
        pk->hk = hk; pk->hk_free = [](void* p) { delete (HostKey<SuiteBn254>*)p; };
        u32 img1[G1::AW], img2[G2::AW];
        auto one1 = [&](size_t off, H1::X* dst) -> int32_t {
            const u32 e = decode_g1(data + off, L.raw, img1);
            if (e) return fail(DVN_ERR_INVALID_ARG, "proving key: a G1 key constant does not decode (error bits %u)", e);
            *dst = H1::load_device_affine(img1);
            return DVN_OK;
        };
        auto one2 = [&](size_t off, H2::X* dst) -> int32_t {
            const u32 e = decode_g2(data + off, L.raw, img2);
            if (e) return fail(DVN_ERR_INVALID_ARG, "proving key: a G2 key constant does not decode (error bits %u)", e);
            *dst = H2::load_device_affine(img2);
            return DVN_OK;
        };
        TRY(one1(L.alpha, &hk->alpha)); TRY(one1(L.beta, &hk->beta)); TRY(one1(L.delta, &hk->delta));
        TRY(one2(L.beta2, &hk->beta2)); TRY(one2(L.delta2, &hk->delta2));
        return DVN_OK;
    };
    int32_t rc = body();
    if (d_bytes) (void)hipFree(d_bytes);
    if (rc != DVN_OK) { std::string keep = tl_error; (void)dvn_groth16_pk_free(ctx, pk); tl_error = keep; return rc; }
    *out = pk;
    return DVN_OK;
}

}  // namespace

int32_t dvn_internal_key_cache_clear(dvn_ctx* ctx) {
    std::vector<std::pair<std::string, dvn_groth16_pk*>> keys;
    { std::lock_guard<std::mutex> lk(ctx->mu); keys.swap(ctx->key_cache); }
    for (auto& kv : keys) { kv.second->cached = false; (void)dvn_groth16_pk_free(ctx, kv.second); }
    return DVN_OK;
}

// Host only: what the file holds, without touching a device (tests on CPU; also a cheap sanity check before a load).
// info: log_n, raw (0 / 1), n_wires, |A|, |B|, |Z|, |K|, |G2.B|, n_commitments, |Basis[0]|, domain flag byte present (0 / 1)
extern "C" int32_t dvn_gnark_pk_scan(const uint8_t* data, size_t len, uint64_t info[11]) try {
    if (!data || !info) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    Layout L; std::string why;
    try {
        if (!scan_any(data, len, &L, &why)) return fail(DVN_ERR_INVALID_ARG, "not a gnark BN254 proving key: %s", why.c_str());
    } catch (...) { return fail(DVN_ERR_OOM, "host allocation failed while reading the key"); }      // nothing is thrown across the C boundary
    info[0] = L.log_n; info[1] = L.raw; info[2] = L.n_wires; info[3] = L.A.count; info[4] = L.B.count; info[5] = L.Z.count; info[6] = L.K.count;
    info[7] = L.B2.count; info[8] = L.n_commitments; info[9] = L.n_commitments ? L.basis[0].count : 0; info[10] = L.precompute_flag;
    return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_groth16_pk_load_bytes(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                             uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    TRY(enter(ctx));
    if (!data || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    try { return load_image(ctx, data, len, n_public, committed_wires, n_committed_wires, out); }
    catch (...) { return fail(DVN_ERR_OOM, "host allocation failed while reading the key"); }
} DVN_CATCH
// path: the artifact file; sha256_hex: its expected content hash (the file's name in the node's cache) or NULL to skip the
// check.  A key with the same hash already resident on this context is returned as is (one upload per process).
extern "C" int32_t dvn_groth16_pk_load_file(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                            const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out) try {
    TRY(enter(ctx));
    if (!path || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(DVN_ERR_INVALID_ARG, "cannot open %s", path);
    struct stat stt;
    if (fstat(fd, &stt) != 0 || stt.st_size <= 0) { close(fd); return fail(DVN_ERR_INVALID_ARG, "cannot stat %s (or it is empty)", path); }
    const size_t len = (size_t)stt.st_size;
    void* map = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);          // streamed in by the kernel, never copied whole on the host
    close(fd);
    if (map == MAP_FAILED) return fail(DVN_ERR_OOM, "cannot map %s", path);
    uint8_t h[32];
    (void)dvn_sha256((const uint8_t*)map, len, h);
    const std::string hex = hex32(h);
    int32_t rc = DVN_OK;
    if (sha256_hex) {
        std::string want(sha256_hex);
        for (char& ch : want) ch = (char)tolower(ch);
        if (want != hex) rc = fail(DVN_ERR_INVALID_ARG, "hash mismatch for %s: expected %s, got %s", path, want.c_str(), hex.c_str());    // circuits/artifacts.go:123-133
    }
    if (rc == DVN_OK) {
        dvn_groth16_pk* hit = nullptr;
        { std::lock_guard<std::mutex> lk(ctx->mu); for (auto& kv : ctx->key_cache) if (kv.first == hex) hit = kv.second; }
        if (hit) *out = hit;
        else {
            try { rc = load_image(ctx, (const uint8_t*)map, len, n_public, committed_wires, n_committed_wires, out); }
            catch (...) { rc = fail(DVN_ERR_OOM, "host allocation failed while reading the key"); }
            if (rc == DVN_OK) {
                // two threads may have loaded the same file at once: the first to get here owns the cache entry, the other
                // hands back the resident key and drops its own copy
                dvn_groth16_pk* loser = nullptr;
                {
                    std::lock_guard<std::mutex> lk(ctx->mu);
                    for (auto& kv : ctx->key_cache) if (kv.first == hex) hit = kv.second;
                    if (hit) { loser = *out; *out = hit; }
                    else { (*out)->cached = true; ctx->key_cache.emplace_back(hex, *out); }
                }
                if (loser) (void)dvn_groth16_pk_free(ctx, loser);
            }
        }
    }
    munmap(map, len);
    return rc;
} DVN_CATCH
// the commitment bases of a loaded key (NULL when the key has no commitment): what the solver's BSB22 hint passes to
// dvn_pedersen_commit.  The handles stay owned by the key.
extern "C" int32_t dvn_groth16_pk_commitment_bases(const dvn_groth16_pk* pk, const dvn_points** basis, const dvn_points** basis_exp_sigma) try {
    if (!pk || !basis || !basis_exp_sigma) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    *basis = pk->Basis; *basis_exp_sigma = pk->BasisSigma;
    return DVN_OK;
} DVN_CATCH
