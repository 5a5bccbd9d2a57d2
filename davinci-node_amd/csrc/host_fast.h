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
typedef Group<OpsFq> HG1;
typedef Group<OpsFq2> HG2;

}  // namespace dvn_host
