// Host-side finishing arithmetic with 4 x 64-bit limbs (Montgomery R = 2^256, i.e. gnark's own form).
//
// The tail of every MSM is a strictly sequential chain -- the MSB-first window combine of gnark-crypto's
// MultiExp (`for j = last..0: acc = 2^c * acc + W_j`): ~256 doublings + 16 additions.
// One GPU wave needs ~5 us per group operation, a CPU core with 64x64->128 multiplies ~0.3 us, so the chain runs
// here.  (The 29-bit device arithmetic also compiles for the host and is what tests/test_host_arith.py checks; this
// file exists only because it is ~4x faster on a CPU.)  Inputs are the per-window sums as the device left them
// (internal form x * 2^261, packed 8 x u32 per coordinate); the output is gnark's Jacobian image with Z = 1.
#pragma once
#include <stdint.h>
#include <string.h>

namespace dvn_host {

typedef uint64_t u64;
typedef unsigned __int128 u128;
struct Fq { u64 l[4]; };

static const u64 P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};   // 2^256 mod p
static const u64 INV = 0x87d20782e4866389ULL;                                                                        // -p^-1 mod 2^64

static inline bool geq(const u64* a, const u64* b) {
    for (int i = 3; i >= 0; --i) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; }
    return true;
}
static inline void sub_p(u64* t) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 d = (u128)t[i] - P[i] - br; t[i] = (u64)d; br = (d >> 64) & 1; } }
static inline Fq add(const Fq& a, const Fq& b) {
    Fq r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (u64)c; c >>= 64; }
    if (c || geq(r.l, P)) sub_p(r.l);
    return r;
}
static inline Fq sub(const Fq& a, const Fq& b) {
    Fq r; u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (u64)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + P[i]; r.l[i] = (u64)c; c >>= 64; } }
    return r;
}
static inline Fq mul(const Fq& a, const Fq& b) {          // CIOS
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        const u64 m = t[0] * INV;
        c = (u128)m * P[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * P[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq(t, P)) sub_p(t);
    Fq r; memcpy(r.l, t, 32); return r;
}
static inline bool is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline Fq zero() { Fq r = {{0, 0, 0, 0}}; return r; }
static inline Fq one() { Fq r; memcpy(r.l, ONE, 32); return r; }
static inline Fq neg(const Fq& a) { return is_zero(a) ? a : sub(zero(), a); }
static inline Fq inv(const Fq& a) {                        // a^(p-2)
    u64 e[4] = {P[0] - 2, P[1], P[2], P[3]};
    Fq acc = one();
    for (int i = 255; i >= 0; --i) { acc = mul(acc, acc); if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, a); }
    return acc;
}
// device word (x * 2^261, packed little-endian) -> Montgomery x * 2^256: multiply by 2^-5, i.e. mont_mul by 2^251
static inline Fq from_device(const uint32_t* w) {
    Fq v; memcpy(v.l, w, 32);
    static const Fq c251 = {{0, 0, 0, 0x0800000000000000ULL}};   // 2^251 < p
    return mul(v, c251);
}

struct Fq2 { Fq a0, a1; };
struct OpsFq {
    typedef Fq F;
    static inline F add(const F& a, const F& b) { return dvn_host::add(a, b); }
    static inline F sub(const F& a, const F& b) { return dvn_host::sub(a, b); }
    static inline F mul(const F& a, const F& b) { return dvn_host::mul(a, b); }
    static inline F sqr(const F& a) { return dvn_host::mul(a, a); }
    static inline F inv(const F& a) { return dvn_host::inv(a); }
    static inline F zero() { return dvn_host::zero(); }
    static inline F one() { return dvn_host::one(); }
    static inline bool is_zero(const F& a) { return dvn_host::is_zero(a); }
    static inline F from_device(const uint32_t* w) { return dvn_host::from_device(w); }
    static inline void store(const F& a, uint64_t* o) { memcpy(o, a.l, 32); }
    static const int WORDS = 8, U64S = 4;
};
struct OpsFq2 {
    typedef Fq2 F;
    static inline F add(const F& a, const F& b) { return F{dvn_host::add(a.a0, b.a0), dvn_host::add(a.a1, b.a1)}; }
    static inline F sub(const F& a, const F& b) { return F{dvn_host::sub(a.a0, b.a0), dvn_host::sub(a.a1, b.a1)}; }
    static inline F mul(const F& a, const F& b) {
        Fq t0 = dvn_host::mul(a.a0, b.a0), t1 = dvn_host::mul(a.a1, b.a1);
        Fq m = dvn_host::mul(dvn_host::add(a.a0, a.a1), dvn_host::add(b.a0, b.a1));
        return F{dvn_host::sub(t0, t1), dvn_host::sub(dvn_host::sub(m, t0), t1)};
    }
    static inline F sqr(const F& a) {
        Fq m = dvn_host::mul(a.a0, a.a1);
        return F{dvn_host::mul(dvn_host::add(a.a0, a.a1), dvn_host::sub(a.a0, a.a1)), dvn_host::add(m, m)};
    }
    static inline F inv(const F& a) {
        Fq d = dvn_host::inv(dvn_host::add(dvn_host::mul(a.a0, a.a0), dvn_host::mul(a.a1, a.a1)));
        return F{dvn_host::mul(a.a0, d), dvn_host::neg(dvn_host::mul(a.a1, d))};
    }
    static inline F zero() { return F{dvn_host::zero(), dvn_host::zero()}; }
    static inline F one() { return F{dvn_host::one(), dvn_host::zero()}; }
    static inline bool is_zero(const F& a) { return dvn_host::is_zero(a.a0) && dvn_host::is_zero(a.a1); }
    static inline F from_device(const uint32_t* w) { return F{dvn_host::from_device(w), dvn_host::from_device(w + 8)}; }
    static inline void store(const F& a, uint64_t* o) { memcpy(o, a.a0.l, 32); memcpy(o + 4, a.a1.l, 32); }
    static const int WORDS = 16, U64S = 8;
};

// extended Jacobian group law (same formulas as bn254_curve.h)
template <class O> struct Group {
    typedef typename O::F F;
    struct X { F x, y, zz, zzz; };
    static inline X infinity() { return X{O::one(), O::one(), O::zero(), O::zero()}; }
    static inline bool is_inf(const X& p) { return O::is_zero(p.zz); }
    static inline X dbl(const X& p) {
        if (is_inf(p)) return p;
        F U = O::add(p.y, p.y), V = O::sqr(U), W = O::mul(U, V), S = O::mul(p.x, V);
        F xx = O::sqr(p.x), M = O::add(O::add(xx, xx), xx);
        F X3 = O::sub(O::sub(O::sqr(M), S), S);
        F Y3 = O::sub(O::mul(M, O::sub(S, X3)), O::mul(W, p.y));
        return X{X3, Y3, O::mul(V, p.zz), O::mul(W, p.zzz)};
    }
    static inline X add(const X& p, const X& q) {
        if (is_inf(q)) return p;
        if (is_inf(p)) return q;
        F U1 = O::mul(p.x, q.zz), U2 = O::mul(q.x, p.zz), S1 = O::mul(p.y, q.zzz), S2 = O::mul(q.y, p.zzz);
        F Pd = O::sub(U2, U1), R = O::sub(S2, S1);
        if (O::is_zero(Pd)) { if (O::is_zero(R)) return dbl(p); return infinity(); }
        F PP = O::sqr(Pd), PPP = O::mul(Pd, PP), Q = O::mul(U1, PP);
        F X3 = O::sub(O::sub(O::sub(O::sqr(R), PPP), Q), Q);
        F Y3 = O::sub(O::mul(R, O::sub(Q, X3)), O::mul(S1, PPP));
        return X{X3, Y3, O::mul(O::mul(p.zz, q.zz), PP), O::mul(O::mul(p.zzz, q.zzz), PPP)};
    }
    static inline X load_device(const uint32_t* w) {       // device xyzz image; all-zero ZZ = infinity
        bool z = true;
        for (int i = 0; i < O::WORDS; ++i) z = z && w[2 * O::WORDS + i] == 0;
        if (z) return infinity();
        return X{O::from_device(w), O::from_device(w + O::WORDS), O::from_device(w + 2 * O::WORDS), O::from_device(w + 3 * O::WORDS)};
    }
    // sum_w 2^(c w) * W_w, MSB first
    static inline X horner(const uint32_t* sums, uint32_t nwin, uint32_t c) {
        X acc = infinity();
        for (int w = (int)nwin - 1; w >= 0; --w) {
            for (uint32_t k = 0; k < c; ++k) acc = dbl(acc);
            acc = add(acc, load_device(sums + (size_t)w * 4 * O::WORDS));
        }
        return acc;
    }
    // [k]p for a 256-bit scalar (8 x u32 little-endian), MSB-first double-and-add
    static inline X mul_scalar(const X& p, const uint32_t k[8]) {
        X acc = infinity();
        for (int b = 255; b >= 0; --b) {
            acc = dbl(acc);
            if ((k[b >> 5] >> (b & 31)) & 1) acc = add(acc, p);
        }
        return acc;
    }
    static inline X mul_scalar_n(const X& p, const uint32_t* k, uint32_t nwords) {
        X acc = infinity();
        for (int b = (int)nwords * 32 - 1; b >= 0; --b) {
            acc = dbl(acc);
            if ((k[b >> 5] >> (b & 31)) & 1) acc = add(acc, p);
        }
        return acc;
    }
    // device affine image (internal form, 2 coordinates; all zero = infinity) -> X with ZZ = ZZZ = 1
    static inline X load_device_affine(const uint32_t* w) {
        bool z = true;
        for (int i = 0; i < 2 * O::WORDS; ++i) z = z && w[i] == 0;
        if (z) return infinity();
        return X{O::from_device(w), O::from_device(w + O::WORDS), O::one(), O::one()};
    }
    static inline X normalize(const X& p) {                 // affine representative (ZZ = ZZZ = 1)
        if (is_inf(p)) return p;
        F zi = O::inv(p.zzz), iz2 = O::sqr(O::mul(zi, p.zz));
        return X{O::mul(p.x, iz2), O::mul(p.y, zi), O::one(), O::one()};
    }
    // gnark Jacobian image: affine normal form with Z = 1, infinity = (1, 1, 0)
    static inline void write_jac(const X& p, uint64_t* out) {
        if (is_inf(p)) { O::store(O::one(), out); O::store(O::one(), out + O::U64S); O::store(O::zero(), out + 2 * O::U64S); return; }
        F zi = O::inv(p.zzz), iz2 = O::sqr(O::mul(zi, p.zz));
        O::store(O::mul(p.x, iz2), out); O::store(O::mul(p.y, zi), out + O::U64S); O::store(O::one(), out + 2 * O::U64S);
    }
};
// ---- launch planning (pure functions: also compiled into the CPU test library) ---------------------------------------------------
// Stage counts of the passes of a 2^log_n transform in EXECUTION order (dif: stages ascending, else descending): ceil(log_n / kmax)
// passes of nearly equal counts.  The pass that holds the transform's last two stages (the last pass of DIF, the first of DIT) gets
// an even count when there is a way -- a swap with an even pass, or one stage handed to a neighbour -- so that those two stages form
// one radix-4 round (the multiplication-free one on plain twiddle tables).  Returns the number of passes (<= 16).
static inline uint32_t ntt_pass_plan(uint32_t log_n, uint32_t kmax, bool dif, uint32_t ks[16]) {
    const uint32_t npass = (log_n + kmax - 1) / kmax;
    const uint32_t base = log_n / npass, rem = log_n % npass;
    for (uint32_t p = 0; p < npass; ++p) ks[p] = base + (p < rem ? 1 : 0);
    if (npass > 1) {
        const uint32_t end = dif ? npass - 1 : 0;
        if (ks[end] & 1) {
            bool done = false;
            for (uint32_t p = 0; p < npass && !done; ++p) if (!(ks[p] & 1)) { const uint32_t t = ks[p]; ks[p] = ks[end]; ks[end] = t; done = true; }
            for (uint32_t p = 0; p < npass && !done; ++p) if (p != end && ks[p] < kmax && ks[end] > 2) { ++ks[p]; --ks[end]; done = true; }
        }
    }
    return npass;
}
// Window size and row count of the expanded table of an n-point set (0 / 1: the set stays plain); see dvn_internal.h expand_plan
// for the measurements behind the numbers.  min_log: smallest expanded set; max_bytes: table size limit.
static inline void expand_plan_pure(size_t n, uint32_t scalar_bits, size_t point_bytes, uint32_t min_log, size_t max_bytes, uint32_t* c, uint32_t* nexp) {
    *c = 0; *nexp = 1;
    if (n < ((size_t)1 << min_log)) return;
    uint32_t cc = n >= ((size_t)1 << 20) ? 20 : n >= ((size_t)1 << 18) ? 17 : n >= ((size_t)1 << 15) ? 16 : 15;
    // a top row of only a few bits (255 = 15 * 17 + 0: nothing but carries; 377 = 22 * 17 + 3) piles a large share of all entries
    // into a handful of buckets: take the next window size whose top row holds at least 8 bits
    while (scalar_bits % cc < 8 && cc < 21) ++cc;
    const uint32_t rows = scalar_bits / cc + 1;
    if ((size_t)rows * n * point_bytes > max_bytes || (size_t)rows * n >= ((size_t)1 << 31)) return;
    *c = cc; *nexp = rows;
}
// ---- host half of the one-bucket-set reduction (msm_kernels.h k_marginal_level / k_marginal_bitsums) -----------------------------
// The device leaves, for the bucket matrix B[hi][lo] (2^kh rows, 2^kl columns, bucket b = hi 2^kl + lo has weight b + 1), plain
// SUBSET sums of its row sums R_hi and column sums C_lo: for every bit j of the row weight hi the sum of the R_hi with that bit set,
// for every bit j of the column weight lo + 1 the sum of the C_lo with that bit set -- each in ceil(count / BITSUM_SPAN) pieces
// (one per workgroup).  Piece order: row bits first (bit-major, pieces of a bit adjacent), then the kl + 1 column bits.
//   sum_b (b + 1) B_b = 2^kl sum_hi hi R_hi + sum_lo (lo + 1) C_lo = sum_j 2^(kl + j) SR_j + sum_j 2^j SC_j
constexpr uint32_t BITSUM_SPAN = 512;
static inline uint32_t bitsum_pieces(uint32_t kh, uint32_t kl) {
    const uint32_t sr = ((1u << kh) + BITSUM_SPAN - 1) / BITSUM_SPAN, sc = ((1u << kl) + BITSUM_SPAN - 1) / BITSUM_SPAN;
    return kh * sr + (kl + 1) * sc;
}
// piece(i) -> the i-th piece as a group element; MSB first, ~2 (kh + kl) group operations
template <class HG, class Piece> static inline typename HG::X bitsum_combine_pieces(Piece piece, uint32_t kh, uint32_t kl) {
    const uint32_t sr = ((1u << kh) + BITSUM_SPAN - 1) / BITSUM_SPAN, sc = ((1u << kl) + BITSUM_SPAN - 1) / BITSUM_SPAN;
    const size_t col0 = (size_t)kh * sr;
    typename HG::X acc = HG::infinity();
    for (int e = (int)(kl + kh); e >= 0; --e) {
        acc = HG::dbl(acc);
        if (e >= (int)kl && e - (int)kl < (int)kh)
            for (uint32_t s = 0; s < sr; ++s) acc = HG::add(acc, piece((size_t)(e - kl) * sr + s));
        if (e <= (int)kl)
            for (uint32_t s = 0; s < sc; ++s) acc = HG::add(acc, piece(col0 + (size_t)e * sc + s));
    }
    return acc;
}
typedef Group<OpsFq> HG1;
typedef Group<OpsFq2> HG2;

// ---- the same finishing arithmetic for the wider fields (BLS12-377 / BLS12-381: 6 limbs, BW6-761: 12 limbs), Montgomery
// R = 2^(64 NW) = gnark's own form.  Cfg is the device field config (29-bit limb tables): the constants are derived from it once.
template <int NW> struct FqN { u64 l[NW]; };
template <class Cfg, int NW> struct OpsFqN {
    typedef FqN<NW> F;
    struct K { u64 p[NW]; u64 n0; F one, conv; };
    static inline bool geq(const u64* a, const u64* b) { for (int i = NW - 1; i >= 0; --i) { if (a[i] != b[i]) return a[i] > b[i]; } return true; }
    static inline void sub_in(u64* a, const u64* b) { u128 br = 0; for (int i = 0; i < NW; ++i) { u128 d = (u128)a[i] - b[i] - br; a[i] = (u64)d; br = (d >> 64) & 1; } }
    static const K& k() {
        static const K c = []() {
            K q; memset(&q, 0, sizeof(q));
            for (int i = 0; i < Cfg::NL; ++i) {                  // modulus from the 29-bit limb table
                const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
                const u64 v = Cfg::p(i);
                if (w < NW) q.p[w] |= v << sh;
                if (sh + 29 > 64 && w + 1 < NW) q.p[w + 1] |= v >> (64 - sh);
            }
            u64 inv = 1;
            for (int i = 0; i < 6; ++i) inv *= 2 - q.p[0] * inv;
            q.n0 = (u64)0 - inv;
            u64 x[NW]; memset(x, 0, sizeof(x)); x[0] = 1;       // 2^(64 NW) mod p by doubling
            for (int i = 0; i < 64 * NW; ++i) {
                const u64 hi = x[NW - 1] >> 63;
                for (int j = NW - 1; j > 0; --j) x[j] = (x[j] << 1) | (x[j - 1] >> 63);
                x[0] <<= 1;
                if (hi || geq(x, q.p)) sub_in(x, q.p);
            }
            memcpy(q.one.l, x, sizeof(x));
            // device word = x * 2^(29 NL); Montgomery product with the integer 2^(2*64 NW - 29 NL) turns it into x * 2^(64 NW)
            const int e = 2 * 64 * NW - 29 * Cfg::NL;
            q.conv.l[e >> 6] = 1ull << (e & 63);
            return q;
        }();
        return c;
    }
    static inline F add(const F& a, const F& b) {
        const K& c = k(); F r; u128 cy = 0;
        for (int i = 0; i < NW; ++i) { cy += (u128)a.l[i] + b.l[i]; r.l[i] = (u64)cy; cy >>= 64; }
        if (cy || geq(r.l, c.p)) sub_in(r.l, c.p);
        return r;
    }
    static inline F sub(const F& a, const F& b) {
        const K& c = k(); F r; u128 br = 0;
        for (int i = 0; i < NW; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (u64)d; br = (d >> 64) & 1; }
        if (br) { u128 cy = 0; for (int i = 0; i < NW; ++i) { cy += (u128)r.l[i] + c.p[i]; r.l[i] = (u64)cy; cy >>= 64; } }
        return r;
    }
    static inline F mul(const F& a, const F& b) {          // CIOS
        const K& c = k();
        u64 t[NW + 2]; memset(t, 0, sizeof(t));
        for (int i = 0; i < NW; ++i) {
            u128 cy = 0;
            for (int j = 0; j < NW; ++j) { cy += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (u64)cy; cy >>= 64; }
            cy += t[NW]; t[NW] = (u64)cy; t[NW + 1] = (u64)(cy >> 64);
            const u64 m = t[0] * c.n0;
            cy = (u128)m * c.p[0] + t[0]; cy >>= 64;
            for (int j = 1; j < NW; ++j) { cy += (u128)m * c.p[j] + t[j]; t[j - 1] = (u64)cy; cy >>= 64; }
            cy += t[NW]; t[NW - 1] = (u64)cy; t[NW] = t[NW + 1] + (u64)(cy >> 64);
        }
        if (t[NW] || geq(t, c.p)) sub_in(t, c.p);
        F r; memcpy(r.l, t, sizeof(r.l)); return r;
    }
    static inline F sqr(const F& a) { return mul(a, a); }
    static inline F zero() { F r; memset(&r, 0, sizeof(r)); return r; }
    static inline F one() { return k().one; }
    static inline bool is_zero(const F& a) { u64 z = 0; for (int i = 0; i < NW; ++i) z |= a.l[i]; return z == 0; }
    static inline F neg(const F& a) { return is_zero(a) ? a : sub(zero(), a); }
    static inline F inv(const F& a) {                       // a^(p-2)
        const K& c = k();
        u64 e[NW]; memcpy(e, c.p, sizeof(e)); e[0] -= 2;
        F acc = c.one;
        for (int i = 64 * NW - 1; i >= 0; --i) { acc = mul(acc, acc); if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, a); }
        return acc;
    }
    static inline F from_device(const uint32_t* w) { F v; memcpy(v.l, w, sizeof(v.l)); return mul(v, k().conv); }
    static inline void store(const F& a, uint64_t* o) { memcpy(o, a.l, sizeof(a.l)); }
    static const int WORDS = 2 * NW, U64S = NW;
};
// quadratic extension Fq[u]/(u^2 + BETA) over it (BLS12-377: BETA = 5, BLS12-381: BETA = 1)
template <class Cfg, int NW, int BETA> struct OpsFq2N {
    typedef OpsFqN<Cfg, NW> B;
    struct F { typename B::F a0, a1; };
    static inline typename B::F mul_beta(const typename B::F& x) {
        typename B::F r = x;
        for (int i = 1; i < BETA; ++i) r = B::add(r, x);
        return r;
    }
    static inline F add(const F& a, const F& b) { return F{B::add(a.a0, b.a0), B::add(a.a1, b.a1)}; }
    static inline F sub(const F& a, const F& b) { return F{B::sub(a.a0, b.a0), B::sub(a.a1, b.a1)}; }
    static inline F mul(const F& a, const F& b) {
        const typename B::F t0 = B::mul(a.a0, b.a0), t1 = B::mul(a.a1, b.a1);
        const typename B::F m = B::mul(B::add(a.a0, a.a1), B::add(b.a0, b.a1));
        return F{B::sub(t0, mul_beta(t1)), B::sub(B::sub(m, t0), t1)};
    }
    static inline F sqr(const F& a) { return mul(a, a); }
    static inline F inv(const F& a) {                       // (a0 - a1 u) / (a0^2 + BETA a1^2)
        const typename B::F d = B::inv(B::add(B::mul(a.a0, a.a0), mul_beta(B::mul(a.a1, a.a1))));
        return F{B::mul(a.a0, d), B::neg(B::mul(a.a1, d))};
    }
    static inline F zero() { return F{B::zero(), B::zero()}; }
    static inline F one() { return F{B::one(), B::zero()}; }
    static inline bool is_zero(const F& a) { return B::is_zero(a.a0) && B::is_zero(a.a1); }
    static inline F from_device(const uint32_t* w) { return F{B::from_device(w), B::from_device(w + 2 * NW)}; }
    static inline void store(const F& a, uint64_t* o) { B::store(a.a0, o); B::store(a.a1, o + NW); }
    static const int WORDS = 4 * NW, U64S = 2 * NW;
};

}  // namespace dvn_host
