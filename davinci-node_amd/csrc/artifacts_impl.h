// gnark proving-key artifact reader, generic over the pairing curve, + the sha256-keyed device-resident key cache.
//
// The node keeps its circuit artifacts as sha256-named files and loads a proving key with
//     pk := newProvingKey(curve); pk.UnsafeReadFrom(file)              /root/reference/circuits/artifacts.go:391-406
// after cmd/circuit-compile wrote it with pk.WriteTo (/root/reference/cmd/circuit-compile/main.go:507-513); at boot it does so for
// the four circuits of a node -- voteverifier (BLS12-377), aggregator (BW6-761), statetransition and results (BN254):
// /root/reference/sequencer/circuit_artifacts.go:39-76.  This file reads the same bytes straight into a device-resident key:
// points are decompressed ON THE DEVICE (one square root per thread) into the internal affine images the MSM kernels use,
// expanded tables are built, and the handle is what the curve's prove_solved entry point takes.
//
// Layout read here (gnark backend/groth16/<curve> ProvingKey.WriteTo / WriteRawTo over gnark-crypto's Encoder; the modules are not
// vendored in the reference tree, so the layout is restated from the published format and pinned only by this repository's own
// writer, oracle/gnark_key_format_py.py -- "parity unpinned" by reference data):
//   fft.Domain       uint64 Cardinality | fr CardinalityInv | fr Generator | fr GeneratorInv | fr FrMultiplicativeGen |
//                    fr FrMultiplicativeGenInv (big-endian canonical, the scalar field's byte width each) | [bool withPrecompute]
//   G1.Alpha, G1.Beta, G1.Delta                       one point each
//   G1.A, G1.B, G1.Z, G1.K                            uint32 big-endian length + points
//   G2.Beta, G2.Delta                                 one point each
//   G2.B                                              uint32 length + points
//   uint64 nbWires | uint64 NbInfinityA | uint64 NbInfinityB | InfinityA | InfinityB   ([]bool: one byte per wire; accepted
//                    with or without a uint32 length in front of each bitmap -- dvn_*_pk_scan reports which)
//   uint32 nbCommitmentKeys, then per key: Basis (uint32 length + points) | BasisExpSigma (uint32 length + points)
// Point encodings (gnark-crypto ecc/<curve>/marshal.go): coordinates are big-endian canonical integers of the base field's byte
// width (32 / 48 / 96), the flag sits in the top bits of the first byte.
//   BN254 (2 spare bits):   00 uncompressed (infinity = all zero), 10 / 11 compressed with the smallest / largest y,
//                           01 compressed infinity
//   BLS12-377, BW6-761 (3): 000 uncompressed, 010 uncompressed infinity, 100 / 101 compressed smallest / largest, 110 compressed
//                           infinity; 001, 011, 111 invalid
// G2 over Fp2 stores X.A1 | X.A0 (| Y.A1 | Y.A0); BW6-761's G2 is a curve over Fp like its G1.  y is "largest" when
// y > (p - 1) / 2, for Fp2 by A1 unless it is zero, then by A0.  UnsafeReadFrom performs no subgroup checks; neither does this
// reader (it does check that every point is on its curve).
#pragma once
#include "groth16_impl.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" int32_t dvn_sha256(const uint8_t* data, size_t len, uint8_t out[32]);

namespace dvn_art {

enum : u32 { ERR_ENCODING = 1, ERR_NOT_ON_CURVE = 2, ERR_NOT_CANONICAL = 4 };
enum Kind : u32 { K_RAW = 0, K_RAW_INF = 1, K_SMALL = 2, K_LARGE = 3, K_CINF = 4, K_BAD = 5 };

// ---- one base-field coordinate on the wire -------------------------------------------------------------------------------
template <class C> struct FpIo {
    static constexpr int EW = C::EW, CB = 4 * C::EW;                       // words / bytes of a coordinate
    static constexpr bool FLAGS3 = (32 * C::EW - C::BITS) >= 3;            // three flag bits where the modulus leaves room
    static constexpr u32 FLAG_MASK = FLAGS3 ? 0x1fu : 0x3fu;               // what remains of the first byte
    static DVN_HD Kind kind(uint8_t b0) {
        if (FLAGS3) {
            switch (b0 >> 5) { case 0: return K_RAW; case 2: return K_RAW_INF; case 4: return K_SMALL; case 5: return K_LARGE; case 6: return K_CINF; default: return K_BAD; }
        }
        switch (b0 >> 6) { case 0: return K_RAW; case 1: return K_CINF; case 2: return K_SMALL; default: return K_LARGE; }
    }
    // CB big-endian bytes -> little-endian words; strip: clear the flag bits of the first byte
    static DVN_HD void words(const uint8_t* b, u32 (&w)[EW], bool strip) {
#pragma unroll
        for (int i = 0; i < EW; ++i) {
            const uint8_t* p = b + 4 * (EW - 1 - i);
            w[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
        }
        if (strip) w[EW - 1] &= (FLAG_MASK << 24) | 0x00ffffffu;
    }
    static DVN_HD bool all_zero(const u32 (&w)[EW]) { u32 z = 0;
#pragma unroll
        for (int i = 0; i < EW; ++i) z |= w[i];
        return z == 0; }
};
template <class C> DVN_HD bool canonical_int(const Fe<C>& x) {
    const Fe<C> c = fe_canonical<C>(x);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d |= c.v[i] ^ x.v[i];
    return d == 0;
}
template <class C> DVN_HD Fe<C> to_int(const Fe<C>& a) { return fe_canonical<C>(Ops<Fe<C>>::mul(a, fe_small<C>(1))); }
// a^e for e given as 29-bit limbs
template <class C> DVN_HD Fe<C> pow_limbs(const Fe<C>& a, const u32 (&e)[C::NL]) {
    typedef Ops<Fe<C>> O;
    Fe<C> acc = fe_one<C>();
    for (int k = C::NL - 1; k >= 0; --k) {
        u32 limb = 0;
#pragma unroll
        for (int j = 0; j < C::NL; ++j) limb = (j == k) ? e[j] : limb;
        for (int bt = LB - 1; bt >= 0; --bt) { acc = O::sqr(acc); if ((limb >> bt) & 1) acc = O::mul(acc, a); }
    }
    return acc;
}
// (p + add) >> sh as 29-bit limbs (add in {-1, +1}, sh < 2 * 29)
template <class C> DVN_HD void p_shifted(int add, int sh, u32 (&e)[C::NL]) {
    u32 t[C::NL];
    i32 carry = add;
#pragma unroll
    for (int k = 0; k < C::NL; ++k) { const i32 v = (i32)C::p(k) + carry; t[k] = (u32)v & LMASK; carry = v >> LB; }
    const int wl = sh / LB, bs = sh % LB;
#pragma unroll
    for (int k = 0; k < C::NL; ++k) {
        const u32 lo = k + wl < C::NL ? t[k + wl < C::NL ? k + wl : 0] : 0u;
        const u32 hi = k + wl + 1 < C::NL ? t[k + wl + 1 < C::NL ? k + wl + 1 : 0] : 0u;
        e[k] = bs ? ((lo >> bs) | (hi << (LB - bs))) & LMASK : lo;
    }
}
// two-adicity of p - 1
template <class C> constexpr int two_adicity() {
    int s = 0;
    for (int k = 0; k < C::NL; ++k) {
        u32 v = C::p(k) - (k == 0 ? 1u : 0u);
        if (v == 0) { s += LB; continue; }
        while (!(v & 1)) { v >>= 1; ++s; }
        return s;
    }
    return s;
}
// Square roots.  p = 3 mod 4 (BN254, BW6-761): a^((p + 1) / 4).  Otherwise (BLS12-377's base field, p - 1 = 2^46 Q) Tonelli-Shanks
// with c = z^Q of a fixed non-residue z handed in by the host.  Returns false when a is not a square.
template <class C> struct SqrtAux { Fe<C> c; };
template <class C> DVN_HD bool fp_sqrt(const Fe<C>& a, const SqrtAux<C>& aux, Fe<C>* out) {
    typedef Ops<Fe<C>> O;
    if constexpr ((C::p(0) & 3u) == 3u) {
        u32 e[C::NL];
        p_shifted<C>(+1, 2, e);
        const Fe<C> y = pow_limbs<C>(a, e);
        *out = y;
        return fe_eq<C>(O::sqr(y), a);
    } else {
        constexpr int S = two_adicity<C>();
        if (O::is_zero(a)) { *out = fe_zero<C>(); return true; }
        u32 e[C::NL];
        p_shifted<C>(-1, S + 1, e);                                    // (Q - 1) / 2
        const Fe<C> t = pow_limbs<C>(a, e);
        Fe<C> x = O::mul(a, t), b = O::mul(x, t), c = aux.c;           // x = a^((Q+1)/2), b = a^Q
        const Fe<C> one = fe_one<C>();
        u32 m = (u32)S;
        for (int guard = 0; guard <= S && !fe_eq<C>(b, one); ++guard) {
            u32 i = 0;
            Fe<C> t2 = b;
            while (i < m && !fe_eq<C>(t2, one)) { t2 = O::sqr(t2); ++i; }
            if (i >= m) return false;                                  // a is not a square
            Fe<C> w = c;
            for (u32 k = 0; k + i + 1 < m; ++k) w = O::sqr(w);
            x = O::mul(x, w); c = O::sqr(w); b = O::mul(b, c); m = i;
        }
        *out = x;
        return fe_eq<C>(O::sqr(x), a);
    }
}
// y > (p - 1) / 2  <=>  2 y > p, y a canonical integer (Element.LexicographicallyLargest)
template <class C> DVN_HD bool fp_lex_largest(const Fe<C>& y_int) {
    const Fe<C> t = fe_add_lazy<C>(y_int, y_int);
    for (int i = C::NL - 1; i >= 0; --i) {
        if (t.v[i] > C::p(i)) return true;
        if (t.v[i] < C::p(i)) return false;
    }
    return false;
}
// Fp2 = Fp[u] / (u^2 + BETA): square root with base-field arithmetic only (the "complex" method): for a = a0 + a1 u,
// s = sqrt(a0^2 + BETA a1^2), x0 = sqrt((a0 +- s) / 2), x1 = a1 / (2 x0).
template <class C> struct Fp2Aux { SqrtAux<C> sq; Fe<C> half, beta_inv; };
template <class C> DVN_HD bool fp2_sqrt(const Fe2<C>& a, const Fp2Aux<C>& aux, Fe2<C>* out) {
    typedef Ops<Fe<C>> O;
    typedef Ops<Fe2<C>> O2;
    const Fe<C> zero = fe_zero<C>();
    if (O::is_zero(a.a1)) {
        Fe<C> r;
        if (fp_sqrt<C>(a.a0, aux.sq, &r)) { *out = Fe2<C>{r, zero}; return true; }
        if (fp_sqrt<C>(O::mul(O::neg(a.a0), aux.beta_inv), aux.sq, &r)) { *out = Fe2<C>{zero, r}; return true; }     // (r u)^2 = -BETA r^2
        return false;
    }
    Fe<C> s;
    if (!fp_sqrt<C>(O::add(O::sqr(a.a0), O2::mul_beta(O::sqr(a.a1))), aux.sq, &s)) return false;
    Fe<C> x0;
    if (!fp_sqrt<C>(O::mul(O::add(a.a0, s), aux.half), aux.sq, &x0)) {
        if (!fp_sqrt<C>(O::mul(O::sub(a.a0, s), aux.half), aux.sq, &x0)) return false;
    }
    const Fe<C> x1 = O::mul(a.a1, O::inv(O::add(x0, x0)));
    const Fe2<C> r{x0, x1};
    const Fe2<C> chk = O2::sqr(r);
    if (!fe_eq<C>(chk.a0, a.a0) || !fe_eq<C>(chk.a1, a.a1)) return false;
    *out = r;
    return true;
}
template <class C> DVN_HD bool fp2_lex_largest(const Fe2<C>& y) {
    const Fe<C> i1 = to_int<C>(y.a1);
    u32 z = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) z |= i1.v[i];
    if (z) return fp_lex_largest<C>(i1);
    return fp_lex_largest<C>(to_int<C>(y.a0));
}

// ---- one point on the wire -> internal affine image ------------------------------------------------------------------------
// curve over Fp: y^2 = x^3 + b
template <class C> struct CurveAux1 { Fe<C> b; SqrtAux<C> sq; };
template <class C> DVN_HD u32 decode_fp_point(const uint8_t* bytes, bool raw_expected, const CurveAux1<C>& cv, u32* out_words) {
    typedef Curve<Fe<C>> Cv;
    typedef Ops<Fe<C>> O;
    typedef FpIo<C> IO;
    const Kind kd = IO::kind(bytes[0]);
    if (kd == K_BAD) return ERR_ENCODING;
    const bool is_raw = kd == K_RAW || kd == K_RAW_INF;
    if (is_raw != raw_expected) return ERR_ENCODING;
    u32 w[IO::EW];
    IO::words(bytes, w, true);
    if (kd == K_CINF || kd == K_RAW_INF) {                                  // infinity: every other bit must be zero
        u32 any = IO::all_zero(w) ? 0u : 1u;
        if (kd == K_RAW_INF) { u32 wy[IO::EW]; IO::words(bytes + IO::CB, wy, false); any |= IO::all_zero(wy) ? 0u : 1u; }
        for (int k = 0; k < Cv::AW; ++k) out_words[k] = 0;
        return any ? ERR_ENCODING : 0;
    }
    const Fe<C> xi = fe_unpack<C>(w);
    if (!canonical_int<C>(xi)) return ERR_NOT_CANONICAL;
    const Fe<C> x = fe_from_canonical<C>(xi);
    const Fe<C> rhs = O::add(O::mul(O::sqr(x), x), cv.b);
    if (kd == K_RAW) {
        u32 wy[IO::EW];
        IO::words(bytes + IO::CB, wy, false);
        const Fe<C> yi = fe_unpack<C>(wy);
        if (!canonical_int<C>(yi)) return ERR_NOT_CANONICAL;
        if (!IO::FLAGS3 && IO::all_zero(w) && IO::all_zero(wy)) { for (int k = 0; k < Cv::AW; ++k) out_words[k] = 0; return 0; }   // two-bit scheme: (0, 0)
        const Fe<C> y = fe_from_canonical<C>(yi);
        if (!fe_eq<C>(O::sqr(y), rhs)) return ERR_NOT_ON_CURVE;
        Cv::store_affine_internal(typename Cv::A{x, y}, out_words);
        return 0;
    }
    Fe<C> y;
    if (!fp_sqrt<C>(rhs, cv.sq, &y)) return ERR_NOT_ON_CURVE;
    if (fp_lex_largest<C>(to_int<C>(y)) != (kd == K_LARGE)) y = O::neg(y);
    Cv::store_affine_internal(typename Cv::A{x, y}, out_words);
    return 0;
}
// curve over Fp2: coordinates as A1 | A0
template <class C> struct CurveAux2 { Fe2<C> b; Fp2Aux<C> sq; };
template <class C> DVN_HD u32 decode_fp2_point(const uint8_t* bytes, bool raw_expected, const CurveAux2<C>& cv, u32* out_words) {
    typedef Curve<Fe2<C>> Cv;
    typedef Ops<Fe2<C>> O2;
    typedef FpIo<C> IO;
    constexpr int CB = IO::CB;
    const Kind kd = IO::kind(bytes[0]);
    if (kd == K_BAD) return ERR_ENCODING;
    const bool is_raw = kd == K_RAW || kd == K_RAW_INF;
    if (is_raw != raw_expected) return ERR_ENCODING;
    u32 w1[IO::EW], w0[IO::EW];
    IO::words(bytes, w1, true); IO::words(bytes + CB, w0, false);           // X.A1 first, then X.A0
    if (kd == K_CINF || kd == K_RAW_INF) {
        u32 any = (IO::all_zero(w1) && IO::all_zero(w0)) ? 0u : 1u;
        if (kd == K_RAW_INF) { u32 a[IO::EW], b2[IO::EW]; IO::words(bytes + 2 * CB, a, false); IO::words(bytes + 3 * CB, b2, false); any |= (IO::all_zero(a) && IO::all_zero(b2)) ? 0u : 1u; }
        for (int k = 0; k < Cv::AW; ++k) out_words[k] = 0;
        return any ? ERR_ENCODING : 0;
    }
    const Fe<C> x1i = fe_unpack<C>(w1), x0i = fe_unpack<C>(w0);
    if (!canonical_int<C>(x1i) || !canonical_int<C>(x0i)) return ERR_NOT_CANONICAL;
    const Fe2<C> x{fe_from_canonical<C>(x0i), fe_from_canonical<C>(x1i)};
    const Fe2<C> rhs = O2::add(O2::mul(O2::sqr(x), x), cv.b);
    if (kd == K_RAW) {
        u32 v1[IO::EW], v0[IO::EW];
        IO::words(bytes + 2 * CB, v1, false); IO::words(bytes + 3 * CB, v0, false);
        const Fe<C> y1i = fe_unpack<C>(v1), y0i = fe_unpack<C>(v0);
        if (!canonical_int<C>(y1i) || !canonical_int<C>(y0i)) return ERR_NOT_CANONICAL;
        if (!IO::FLAGS3 && IO::all_zero(w1) && IO::all_zero(w0) && IO::all_zero(v1) && IO::all_zero(v0)) { for (int k = 0; k < Cv::AW; ++k) out_words[k] = 0; return 0; }
        const Fe2<C> y{fe_from_canonical<C>(y0i), fe_from_canonical<C>(y1i)};
        const Fe2<C> yy = O2::sqr(y);
        if (!fe_eq<C>(yy.a0, rhs.a0) || !fe_eq<C>(yy.a1, rhs.a1)) return ERR_NOT_ON_CURVE;
        Cv::store_affine_internal(typename Cv::A{x, y}, out_words);
        return 0;
    }
    Fe2<C> y;
    if (!fp2_sqrt<C>(rhs, cv.sq, &y)) return ERR_NOT_ON_CURVE;
    if (fp2_lex_largest<C>(y) != (kd == K_LARGE)) y = O2::neg(y);
    Cv::store_affine_internal(typename Cv::A{x, y}, out_words);
    return 0;
}

// A group of a suite on the wire: which decoder, which constants, how many bytes.  Cv = Curve<Fe<C>> or Curve<Fe2<C>>.
template <class Cv> struct GroupCodec;
template <class C> struct GroupCodec<Curve<Fe<C>>> {
    typedef CurveAux1<C> Aux;
    static constexpr size_t COMPRESSED = FpIo<C>::CB, RAW = 2 * FpIo<C>::CB;
    static DVN_HD u32 decode(const uint8_t* b, bool raw, const Aux& a, u32* out) { return decode_fp_point<C>(b, raw, a, out); }
    static DVN_HD bool first_is_raw(uint8_t b0) { const Kind k = FpIo<C>::kind(b0); return k == K_RAW || k == K_RAW_INF; }
};
template <class C> struct GroupCodec<Curve<Fe2<C>>> {
    typedef CurveAux2<C> Aux;
    static constexpr size_t COMPRESSED = 2 * FpIo<C>::CB, RAW = 4 * FpIo<C>::CB;
    static DVN_HD u32 decode(const uint8_t* b, bool raw, const Aux& a, u32* out) { return decode_fp2_point<C>(b, raw, a, out); }
    static DVN_HD bool first_is_raw(uint8_t b0) { const Kind k = FpIo<C>::kind(b0); return k == K_RAW || k == K_RAW_INF; }
};
template <class Cv> __global__ void k_decode_points(const uint8_t* __restrict__ in, u32* __restrict__ pts, u32 n, u32 stride, u32 raw,
                                                    typename GroupCodec<Cv>::Aux aux, u32* __restrict__ err) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[Cv::AW];
    const u32 e = GroupCodec<Cv>::decode(in + (size_t)stride * i, raw != 0, aux, w);
    if (e) { atomicOr(err, e); return; }
    store_words16(pts + (size_t)i * Cv::AW, w, Cv::AW);
}

// host: the constants a curve's decoder needs.  z: a quadratic non-residue of the base field (only read when p = 1 mod 4).
template <class C> static SqrtAux<C> make_sqrt_aux(u32 z) {
    SqrtAux<C> a; a.c = fe_one<C>();
    if constexpr ((C::p(0) & 3u) != 3u) {
        constexpr int S = two_adicity<C>();
        u32 e[C::NL];
        p_shifted<C>(-1, S, e);                                        // Q
        a.c = pow_limbs<C>(fe_from_canonical<C>(fe_small<C>(z)), e);
    }
    return a;
}
template <class C> static CurveAux1<C> make_aux1(const Fe<C>& b, u32 z) { return CurveAux1<C>{b, make_sqrt_aux<C>(z)}; }
template <class C> static CurveAux2<C> make_aux2(const Fe2<C>& b, u32 z) {
    typedef Ops<Fe<C>> O;
    CurveAux2<C> a; a.b = b; a.sq.sq = make_sqrt_aux<C>(z);
    a.sq.half = O::inv(fe_from_canonical<C>(fe_small<C>(2)));
    a.sq.beta_inv = O::inv(fe_from_canonical<C>(fe_small<C>((u32)Beta<C>::value)));
    return a;
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
    uint32_t log_n = 0; bool precompute_flag = false, raw = false, bitmap_prefix = false;
    size_t alpha = 0, beta = 0, delta = 0, beta2 = 0, delta2 = 0;             // offsets of the single points
    Section A, B, Z, K, B2;
    uint64_t n_wires = 0, n_inf_a = 0, n_inf_b = 0;
    size_t inf_a = 0, inf_b = 0;
    uint32_t n_commitments = 0;
    std::vector<Section> basis, basis_sigma;
    size_t end = 0;
};
// The Domain header must be consistent with itself: a misaligned header (e.g. a wrong guess about the flag byte) is rejected here
// instead of being trial-parsed further.  Checks: CardinalityInv * Cardinality = 1, Generator^Cardinality = 1 and
// Generator^(Cardinality / 2) = -1 (a primitive root), Generator * GeneratorInv = 1, FrMultiplicativeGen = the curve's coset
// shift, FrMultiplicativeGen * FrMultiplicativeGenInv = 1.
template <class S> static bool domain_fields_agree(const uint8_t* hdr, uint64_t card, u32 log_n, u32 coset_gen, std::string* why) {
    typedef FpIo<S> IO;
    typedef Ops<Fe<S>> O;
    Fe<S> f[5];
    for (int i = 0; i < 5; ++i) {
        u32 w[IO::EW];
        IO::words(hdr + 8 + (size_t)i * IO::CB, w, false);
        const Fe<S> v = fe_unpack<S>(w);
        if (!canonical_int<S>(v)) { *why = "a domain field is not below the scalar modulus"; return false; }
        f[i] = fe_from_canonical<S>(v);
    }
    const Fe<S> one = fe_one<S>();
    u32 cw[S::EW] = {(u32)card, (u32)(card >> 32)};
    const Fe<S> cardf = fe_from_canonical<S>(fe_unpack<S>(cw));
    if (!fe_eq<S>(O::mul(f[0], cardf), one)) { *why = "CardinalityInv * Cardinality != 1"; return false; }
    Fe<S> g = f[1];
    for (u32 i = 0; i + 1 < log_n; ++i) g = O::sqr(g);
    if (!fe_eq<S>(g, O::neg(one)) || !fe_eq<S>(O::sqr(g), one)) { *why = "Generator is not a primitive root of unity of order Cardinality"; return false; }
    if (!fe_eq<S>(O::mul(f[1], f[2]), one)) { *why = "Generator * GeneratorInv != 1"; return false; }
    if (!fe_eq<S>(f[3], fe_from_canonical<S>(fe_small<S>(coset_gen)))) { *why = "FrMultiplicativeGen is not the curve's coset generator"; return false; }
    if (!fe_eq<S>(O::mul(f[3], f[4]), one)) { *why = "FrMultiplicativeGen * FrMultiplicativeGenInv != 1"; return false; }
    return true;
}

// what a suite's key file is made of
template <class SU> struct KeyTraits;          // G1_B (curve constants), coset generator, non-residue: specialised next to each suite

// walk the sections; with_flag: the Domain carries the 1-byte withPrecompute bool; prefixed: each []bool carries a uint32 length
template <class SU> static bool scan(const uint8_t* d, size_t len, bool with_flag, bool prefixed, Layout* L, std::string* why) {
    typedef typename SU::FrC S;
    typedef GroupCodec<typename SU::G1T> C1;
    typedef GroupCodec<typename SU::G2T> C2;
    constexpr size_t FRB = FpIo<S>::CB;
    Cursor c(d, len);
    const uint64_t card = c.u64be();
    if (!c.ok || card < 2 || (card & (card - 1)) || card > ((uint64_t)1 << 27)) { *why = "domain cardinality is not a power of two in [2, 2^27]"; return false; }
    L->log_n = 0; while (((uint64_t)1 << L->log_n) < card) ++L->log_n;
    c.take(5 * FRB);
    L->precompute_flag = with_flag; L->bitmap_prefix = prefixed;
    if (with_flag) { const uint8_t* f = c.take(1); if (!f || *f > 1) { *why = "domain flag byte is not a bool"; return false; } }
    if (!c.ok) { *why = "truncated domain header"; return false; }
    if (!domain_fields_agree<S>(d, card, L->log_n, KeyTraits<SU>::COSET_GEN, why)) return false;
    const uint8_t* first = d + c.off;
    if (c.off >= len) { *why = "truncated"; return false; }
    L->raw = C1::first_is_raw(first[0]);
    const size_t g1 = L->raw ? C1::RAW : C1::COMPRESSED, g2 = L->raw ? C2::RAW : C2::COMPRESSED;
    auto single = [&](size_t* where, size_t sz) { *where = c.off; c.take(sz); };
    auto slice = [&](Section* s, size_t sz) { s->count = c.u32be(); s->off = c.off; s->raw = L->raw; c.take((size_t)s->count * sz); };
    single(&L->alpha, g1); single(&L->beta, g1); single(&L->delta, g1);
    slice(&L->A, g1); slice(&L->B, g1); slice(&L->Z, g1); slice(&L->K, g1);
    single(&L->beta2, g2); single(&L->delta2, g2);
    slice(&L->B2, g2);
    L->n_wires = c.u64be(); L->n_inf_a = c.u64be(); L->n_inf_b = c.u64be();
    if (!c.ok || L->n_wires > ((uint64_t)1 << 31)) { *why = "truncated key or implausible wire count"; return false; }
    if (prefixed && c.u32be() != L->n_wires) { *why = "InfinityA length prefix differs from nbWires"; return false; }
    L->inf_a = c.off; c.take((size_t)L->n_wires);
    if (prefixed && c.u32be() != L->n_wires) { *why = "InfinityB length prefix differs from nbWires"; return false; }
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
template <class SU> static bool scan_any(const uint8_t* d, size_t len, Layout* L, std::string* why) {
    std::string w[4];
    // gnark-crypto >= 0.12 writes the withPrecompute bool; []bool goes through encoding/binary (no length) in the pinned versions
    const bool flag[4] = {true, false, true, false}, pre[4] = {false, false, true, true};
    for (int k = 0; k < 4; ++k) if (scan<SU>(d, len, flag[k], pre[k], L, &w[k])) return true;
    *why = w[0] + " (domain flag byte, plain bitmaps) / " + w[1] + " (no flag byte, plain bitmaps) / " + w[2] + " (flag byte, length-prefixed bitmaps) / " + w[3] + " (no flag byte, length-prefixed bitmaps)";
    return false;
}

template <class SU, class Cv> static int32_t decode_slice(dvn_ctx* ctx, const uint8_t* d_bytes, const Section& s, int curve,
                                                           const typename GroupCodec<Cv>::Aux& aux, dvn_points** out) {
    typedef GroupCodec<Cv> GC;
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = curve; p->n = s.count; p->suite = SU::SUITE;
    expand_plan(s.count, SU::FrC::BITS, (size_t)Cv::AW * 4, &p->expand_c, &p->nexp);
    const size_t point_bytes = s.raw ? GC::RAW : GC::COMPRESSED;
    u32* err = nullptr;
    hipError_t e = hipMalloc((void**)&p->d, s.count ? (size_t)s.count * Cv::AW * 4 * p->nexp : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&err, 4);
    if (e == hipSuccess) e = hipMemsetAsync(err, 0, 4, ctx->stream);
    u32 herr = 0;
    if (e == hipSuccess && s.count) {
        k_decode_points<Cv><<<(s.count + 63) / 64, 64, 0, ctx->stream>>>(d_bytes + s.off, p->d, s.count, (u32)point_bytes, s.raw ? 1u : 0u, aux, err);
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

static inline std::string hex32(const uint8_t h[32]) {
    static const char* d = "0123456789abcdef";
    std::string s(64, '0');
    for (int i = 0; i < 32; ++i) { s[2 * i] = d[h[i] >> 4]; s[2 * i + 1] = d[h[i] & 15]; }
    return s;
}

template <class SU> static int32_t load_image(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed, uint32_t n_committed,
                                              dvn_groth16_pk** out) {
    typedef typename SU::FrC S;
    typedef typename SU::G1T G1;
    typedef typename SU::G2T G2;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    constexpr size_t EB = (size_t)S::EW * 4;
    Layout L; std::string why;
    if (!scan_any<SU>(data, len, &L, &why)) return fail(DVN_ERR_INVALID_ARG, "not a gnark %s proving key: %s", KeyTraits<SU>::NAME, why.c_str());
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
    pk->suite = SU::SUITE; pk->imported = true; pk->device = ctx->device;
    pk->scalar_bytes = S::EW * 4; pk->g1_bytes = G1::AW * 4; pk->g2_bytes = G2::AW * 4;
    pk->log_n = L.log_n; pk->n = n; pk->m = m; pk->n_public = n_public; pk->n_constraints = n;
    pk->nA = L.A.count; pk->nB = L.B.count; pk->nK = L.K.count; pk->k_first = m - L.K.count;
    pk->load_n_public = n_public; pk->load_committed.assign(committed, committed + n_committed);
    uint8_t* d_bytes = nullptr;
    const auto aux1 = KeyTraits<SU>::aux_g1();
    const auto aux2 = KeyTraits<SU>::aux_g2();
    auto body = [&]() -> int32_t {
        TRY(SU::domain_create(ctx, L.log_n, &pk->dom));
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIP_TRY(hipMalloc((void**)&d_bytes, len));
        HIP_TRY(hipMemcpy(d_bytes, data, len, hipMemcpyHostToDevice));
        TRY((decode_slice<SU, G1>(ctx, d_bytes, L.A, 1, aux1, &pk->A)));
        TRY((decode_slice<SU, G1>(ctx, d_bytes, L.B, 1, aux1, &pk->B)));
        TRY((decode_slice<SU, G1>(ctx, d_bytes, L.Z, 1, aux1, &pk->Z)));
        TRY((decode_slice<SU, G1>(ctx, d_bytes, L.K, 1, aux1, &pk->K)));
        TRY((decode_slice<SU, G2>(ctx, d_bytes, L.B2, 2, aux2, &pk->B2)));
        if (L.n_commitments == 1) {
            TRY((decode_slice<SU, G1>(ctx, d_bytes, L.basis[0], 1, aux1, &pk->Basis)));
            TRY((decode_slice<SU, G1>(ctx, d_bytes, L.basis_sigma[0], 1, aux1, &pk->BasisSigma)));
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
        HostKey<SU>* hk = new HostKey<SU>();
        pk->hk = hk; pk->hk_free = [](void* p) { delete (HostKey<SU>*)p; };
        u32 img1[G1::AW], img2[G2::AW];
        auto one1 = [&](size_t off, typename H1::X* dst) -> int32_t {
            const u32 e = GroupCodec<G1>::decode(data + off, L.raw, aux1, img1);
            if (e) return fail(DVN_ERR_INVALID_ARG, "proving key: a G1 key constant does not decode (error bits %u)", e);
            *dst = H1::load_device_affine(img1);
            return DVN_OK;
        };
        auto one2 = [&](size_t off, typename H2::X* dst) -> int32_t {
            const u32 e = GroupCodec<G2>::decode(data + off, L.raw, aux2, img2);
            if (e) return fail(DVN_ERR_INVALID_ARG, "proving key: a G2 key constant does not decode (error bits %u)", e);
            *dst = H2::load_device_affine(img2);
            return DVN_OK;
        };
        TRY(one1(L.alpha, &hk->alpha)); TRY(one1(L.beta, &hk->beta)); TRY(one1(L.delta, &hk->delta));
        TRY(one2(L.beta2, &hk->beta2)); TRY(one2(L.delta2, &hk->delta2));
        return DVN_OK;
    };
    int32_t rc;
    try { rc = body(); }
    catch (...) { rc = dvn_guard_fail(); }                     // the key and the staged bytes are released below, then the code returns
    if (d_bytes) (void)hipFree(d_bytes);
    if (rc != DVN_OK) { std::string keep = tl_error; (void)dvn_groth16_pk_free(ctx, pk); tl_error = keep; return rc; }
    *out = pk;
    return DVN_OK;
}

template <class SU> static int32_t pk_scan_api(const uint8_t* data, size_t len, uint64_t info[12]) {
    if (!data || !info) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    Layout L; std::string why;
    if (!scan_any<SU>(data, len, &L, &why)) return fail(DVN_ERR_INVALID_ARG, "not a gnark %s proving key: %s", KeyTraits<SU>::NAME, why.c_str());
    info[0] = L.log_n; info[1] = L.raw; info[2] = L.n_wires; info[3] = L.A.count; info[4] = L.B.count; info[5] = L.Z.count; info[6] = L.K.count;
    info[7] = L.B2.count; info[8] = L.n_commitments; info[9] = L.n_commitments ? L.basis[0].count : 0; info[10] = L.precompute_flag; info[11] = L.bitmap_prefix;
    return DVN_OK;
}
template <class SU> static int32_t pk_load_bytes_api(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                                     uint32_t n_committed_wires, dvn_groth16_pk** out) {
    TRY(enter(ctx));
    if (!data || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    return load_image<SU>(ctx, data, len, n_public, committed_wires, n_committed_wires, out);
}
// a cached key answers a load only when it was loaded for the same public-wire count and committed wires
static inline bool same_load_args(const dvn_groth16_pk* pk, uint32_t n_public, const uint32_t* committed, uint32_t n_committed) {
    if (pk->load_n_public != n_public || pk->load_committed.size() != n_committed) return false;
    for (uint32_t i = 0; i < n_committed; ++i) if (pk->load_committed[i] != committed[i]) return false;
    return true;
}
// path: the artifact file; sha256_hex: its expected content hash (the file's name in the node's cache) or NULL to skip the
// check.  A key with the same hash already resident on this context is returned as is (one upload per process).
template <class SU> static int32_t pk_load_file_api(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                                    const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out) {
    TRY(enter(ctx));
    if (!path || !out || (!committed_wires && n_committed_wires)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(DVN_ERR_INVALID_ARG, "cannot open %s", path);
    struct stat stt;
    if (fstat(fd, &stt) != 0 || stt.st_size <= 0) { close(fd); return fail(DVN_ERR_INVALID_ARG, "cannot stat %s (or it is empty)", path); }
    const size_t len = (size_t)stt.st_size;
    void* map = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);          // streamed in by the kernel, never copied whole on the host
    close(fd);
    if (map == MAP_FAILED) return fail(DVN_ERR_OOM, "cannot map %s", path);
    int32_t rc = DVN_OK;
    try {
        uint8_t h[32];
        (void)dvn_sha256((const uint8_t*)map, len, h);
        const std::string hex = hex32(h);
        if (sha256_hex) {
            std::string want(sha256_hex);
            for (char& ch : want) ch = (char)tolower(ch);
            if (want != hex) rc = fail(DVN_ERR_INVALID_ARG, "hash mismatch for %s: expected %s, got %s", path, want.c_str(), hex.c_str());    // circuits/artifacts.go:123-133
        }
        if (rc == DVN_OK) {
            dvn_groth16_pk* hit = nullptr;
            { std::lock_guard<std::mutex> lk(ctx->mu); for (auto& kv : ctx->key_cache) if (kv.first == hex) hit = kv.second; }
            if (hit) {
                if (hit->suite != SU::SUITE) rc = fail(DVN_ERR_INVALID_ARG, "the resident key with this hash belongs to another curve");
                else if (!same_load_args(hit, n_public, committed_wires, n_committed_wires))
                    rc = fail(DVN_ERR_INVALID_ARG, "the resident key with this hash was loaded with n_public = %u and %zu committed wires; this call names others (evict it with dvn_groth16_pk_evict first)", hit->load_n_public, hit->load_committed.size());
                else *out = hit;
            } else {
                rc = load_image<SU>(ctx, (const uint8_t*)map, len, n_public, committed_wires, n_committed_wires, out);
                if (rc == DVN_OK) {
                    // two threads may have loaded the same file at once: the first to get here owns the cache entry, the other
                    // hands back the resident key and drops its own copy
                    dvn_groth16_pk* loser = nullptr;
                    {
                        std::lock_guard<std::mutex> lk(ctx->mu);
                        for (auto& kv : ctx->key_cache) if (kv.first == hex) hit = kv.second;
                        if (hit && hit->suite == SU::SUITE && same_load_args(hit, n_public, committed_wires, n_committed_wires)) { loser = *out; *out = hit; }
                        else if (!hit) { (*out)->cached = true; ctx->key_cache.emplace_back(hex, *out); }
                    }
                    if (loser) (void)dvn_groth16_pk_free(ctx, loser);
                }
            }
        }
    } catch (...) { rc = dvn_guard_fail(); }
    munmap(map, len);
    return rc;
}

}  // namespace dvn_art
