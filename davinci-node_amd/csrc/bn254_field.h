// BN254 Fp / Fr arithmetic for CDNA4 (gfx950), shared by device kernels and the host-side combine code.
//
// Replaces, on the device, gnark-crypto's ecc/bn254/fp and ecc/bn254/fr Element arithmetic that
// groth16.Prove (called at /root/reference/prover/prover_cpu.go:37, prover_gpu.go:33) spends its time in.
//
// Representation (MI355X-first):
//   * 9 limbs x 29 bits in 32-bit VGPRs ("unsaturated"), Montgomery radix R' = 2^261.
//     gfx950's only fast wide multiplier is v_mad_u64_u32 (32x32+64 -> 64, measured 4 cycles per wave64);
//     32-bit add-with-carry chains need s_nop hazard padding on this target, and a saturated 8x32 CIOS
//     compiles to 2.3 v_mov per multiply-add to build zero-extended pairs.  With 29-bit limbs every
//     partial product is < 2^58, so a column of up to 18 products accumulates in one 64-bit register pair
//     with no carry handling at all: one v_mad_u64_u32 per partial product, nothing else in the hot loop.
//   * Values are kept in [0, 2p) with limbs < 2^29 ("weak form").  mul output is < 1.03 p for inputs < 2p
//     because R' / p > 2^7.
//   * The external format is gnark's: 4 x u64 little-endian limbs, Montgomery R = 2^256.
//     ext -> internal multiplies by 2^5 (one Montgomery multiplication by 2^266 mod p), internal -> ext by
//     2^-5.  Linear maps (NTT) skip both: an external Montgomery word is read as the internal form of
//     x * 2^-5 and the scale factor commutes with the transform.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DVN_HD __host__ __device__ __forceinline__
#define DVN_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define DVN_HD inline
#define DVN_NOINLINE __attribute__((noinline))
#endif

namespace dvn {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

constexpr int LB = 29;                // bits per limb
constexpr u32 LMASK = (1u << LB) - 1;

// Constant tables as constexpr functions of a compile-time index: after unrolling they become literals.
#define DVN_TABLE(name, ...)                                             \
    static DVN_HD constexpr u32 name(int i) {                            \
        constexpr u32 t[] = {__VA_ARGS__};                               \
        return t[i];                                                     \
    }

// FpCfg / FrCfg (BN254), Fp377Cfg / Fr377Cfg (BLS12-377): NL limbs, EW external u32 words, p, 2p, 4p, 8p, R' mod p, ...
#include "field_constants.h"

template <class C>
struct Fe {
    u32 v[C::NL];
};
template <class C> DVN_HD Fe<C> fe_zero() { Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = 0;
    return r; }
template <class C> DVN_HD Fe<C> fe_one() { Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = C::one(i);
    return r; }

// Signed carry propagation: limbs may be any int32; result limbs 0..7 in [0, 2^29), limb 8 keeps the sign.
template <class C> DVN_HD void fe_carry(i32 (&t)[C::NL]) {
#pragma unroll
    for (int i = 0; i < C::NL - 1; ++i) {
        i32 c = t[i] >> LB;            // arithmetic shift: floor division
        t[i] &= (i32)LMASK;
        t[i + 1] += c;
    }
}

// r = a + b, weak form in / weak form out
template <class C> DVN_HD Fe<C> fe_add(const Fe<C>& a, const Fe<C>& b) {
    i32 s[C::NL], d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) { s[i] = (i32)(a.v[i] + b.v[i]); d[i] = s[i] - (i32)C::p2(i); }
    fe_carry<C>(s); fe_carry<C>(d);
    const bool neg = d[C::NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = (u32)(neg ? s[i] : d[i]);
    return r;
}
// r = a - b
template <class C> DVN_HD Fe<C> fe_sub(const Fe<C>& a, const Fe<C>& b) {
    i32 s[C::NL], d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) { d[i] = (i32)a.v[i] - (i32)b.v[i]; s[i] = d[i] + (i32)C::p2(i); }
    fe_carry<C>(s); fe_carry<C>(d);
    const bool neg = d[C::NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = (u32)(neg ? s[i] : d[i]);
    return r;
}
template <class C> DVN_HD Fe<C> fe_dbl(const Fe<C>& a) { return fe_add<C>(a, a); }
template <class C> DVN_HD Fe<C> fe_neg(const Fe<C>& a) { return fe_sub<C>(fe_zero<C>(), a); }

// Montgomery product a*b/2^261 mod p.  Inputs: limbs < 2^30, values < 13p.  Output weak form (< 1.03p .. 2p).
template <class C> DVN_HD Fe<C> fe_mul(const Fe<C>& a, const Fe<C>& b) {
    constexpr int NL = C::NL;
    u64 T[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) T[k] = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i)
#pragma unroll
        for (int j = 0; j < NL; ++j) T[i + j] += (u64)a.v[i] * b.v[j];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const u32 m = ((u32)T[k] * C::inv) & LMASK;
#pragma unroll
        for (int j = 0; j < NL; ++j) T[k + j] += (u64)m * C::p(j);
        T[k + 1] += T[k] >> LB;
    }
    Fe<C> r;
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) { T[k + 1] += T[k] >> LB; r.v[k - NL] = (u32)T[k] & LMASK; }
    r.v[NL - 1] = (u32)T[2 * NL - 1];
    return r;
}
// Dedicated squaring: 45 distinct partial products instead of 81.
template <class C> DVN_HD Fe<C> fe_sqr(const Fe<C>& a) {
    constexpr int NL = C::NL;
    u64 T[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) T[k] = 0;
    u32 a2[NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) a2[i] = a.v[i] << 1;          // < 2^31 for limbs < 2^30
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        T[2 * i] += (u64)a.v[i] * a.v[i];
#pragma unroll
        for (int j = i + 1; j < NL; ++j) T[i + j] += (u64)a2[i] * a.v[j];
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const u32 m = ((u32)T[k] * C::inv) & LMASK;
#pragma unroll
        for (int j = 0; j < NL; ++j) T[k + j] += (u64)m * C::p(j);
        T[k + 1] += T[k] >> LB;
    }
    Fe<C> r;
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) { T[k + 1] += T[k] >> LB; r.v[k - NL] = (u32)T[k] & LMASK; }
    r.v[NL - 1] = (u32)T[2 * NL - 1];
    return r;
}

template <class C> DVN_HD bool fe_is_zero(const Fe<C>& a);

// ---- lazily reduced forms (hot loop of k_accumulate) --------------------------------------------------------
// "Loose" elements have normalised limbs (limbs 0..7 < 2^29) but a value anywhere below 13p; fe_mul / fe_sqr accept them
// (output < p (1 + A B / 169) for inputs < A p, B p because 2^261 / p > 169) so most additive steps need one carry
// pass and no conditional correction.
template <class C, int K> DVN_HD constexpr u32 fe_kp(int i) { return K == 2 ? C::p2(i) : K == 4 ? C::p4(i) : C::p8(i); }
// a + K p - b  (K in {2, 4, 8}; requires b < K p): value < a + K p, limbs normalised
template <class C, int K> DVN_HD Fe<C> fe_sub_lazy(const Fe<C>& a, const Fe<C>& b) {
    i32 d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d[i] = (i32)a.v[i] + (i32)fe_kp<C, K>(i) - (i32)b.v[i];
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// K p - b
template <class C, int K> DVN_HD Fe<C> fe_neg_lazy(const Fe<C>& b) {
    i32 d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d[i] = (i32)fe_kp<C, K>(i) - (i32)b.v[i];
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// a + b with one carry pass, no reduction: value a + b, limbs normalised
template <class C> DVN_HD Fe<C> fe_add_lazy(const Fe<C>& a, const Fe<C>& b) {
    i32 d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d[i] = (i32)(a.v[i] + b.v[i]);
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// x >= K p ? x - K p : x   (x loose with normalised limbs)
template <class C, int K> DVN_HD Fe<C> fe_csub(const Fe<C>& x) {
    i32 d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d[i] = (i32)x.v[i] - (i32)fe_kp<C, K>(i);
    fe_carry<C>(d);
    const bool neg = d[C::NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = neg ? x.v[i] : (u32)d[i];
    return r;
}
// neg ? (4p - s - y) : (s + 2p - y)  for s, y < 2p: the lazily reduced difference of fe_sub_lazy<2> with the sign of s folded in
// (one carry pass; no separate negation of s).  Value below 4p either way.
template <class C> DVN_HD Fe<C> fe_sub_lazy_cneg(const Fe<C>& s, const Fe<C>& y, bool neg) {
    i32 d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const i32 pos = (i32)s.v[i] + (i32)C::p2(i), ng = (i32)C::p4(i) - (i32)s.v[i];
        d[i] = (neg ? ng : pos) - (i32)y.v[i];
    }
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// a + 6p - b - 2c with one carry pass (requires b + 2c <= 6p): value below a + 6p
template <class C> DVN_HD Fe<C> fe_sub2_lazy6(const Fe<C>& a, const Fe<C>& b, const Fe<C>& c) {
    i32 d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d[i] = (i32)a.v[i] + (i32)(C::p2(i) + C::p4(i)) - (i32)b.v[i] - (i32)(c.v[i] << 1);
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// a + b without carry propagation: limbs < 2^30; only valid as the subtrahend of fe_sub_lazy or a fe_mul operand
template <class C> DVN_HD Fe<C> fe_add_nocarry(const Fe<C>& a, const Fe<C>& b) {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// ---- carry-free forms (NTT butterflies) ---------------------------------------------------------------------------
// limb i of K p with limbs 0..NL-2 normalised; folds to a literal once the caller's limb loop is unrolled
template <class C, int K> DVN_HD constexpr u32 fe_kp_limb(int i) {
    u64 carry = 0;
    u32 out = 0;
    for (int j = 0; j <= i; ++j) {
        const u64 v = (u64)C::p(j) * (u32)K + carry;
        out = j < C::NL - 1 ? (u32)(v & LMASK) : (u32)v;
        carry = v >> LB;
    }
    return out;
}
// a + K p - b with NO carry pass.  Each limb of K p below the top one is raised by 2^SH and the limb above it lowered by
// 2^(SH-29) (the same amount), so no limb-wise difference goes negative while b's limbs are < 2^SH and b's top limb is
// <= (K p)'s top limb - 2^(SH-29).  Result: value a + K p - b, limbs < (a's limb bound) + 2^SH + 2^29 -- not normalised, but a
// valid FIRST operand of fe_mul against a normalised second operand as long as fe_mul_raw_fits<C>() accepts that bound.
template <class C, int K, int SH> DVN_HD Fe<C> fe_sub_raw(const Fe<C>& a, const Fe<C>& b) {
    static_assert(SH >= LB && SH <= 31, "offset pattern");
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const u32 k = fe_kp_limb<C, K>(i) + (i < C::NL - 1 ? (1u << SH) : 0u) - (i > 0 ? (1u << (SH - LB)) : 0u);
        r.v[i] = a.v[i] + k - b.v[i];
    }
    return r;
}
// the 64-bit columns of fe_mul hold NL products of a raw limb (< max_limb) with a normalised one plus the reduction's NL
// products of two normalised limbs plus a carry-in
template <class C> DVN_HD constexpr bool fe_mul_raw_fits(u64 max_limb) {
    const unsigned __int128 cols = (unsigned __int128)C::NL * max_limb * LMASK + (unsigned __int128)C::NL * LMASK * LMASK + ((unsigned __int128)1 << 36);
    return (cols >> 64) == 0;
}
// fe_mul_add(a, b, c, d) with a raw a (limbs below 3 * 2^29, as fe_sub_raw<.., .., 29> of normalised values gives) and b, c, d
// normalised: NL raw products + NL normalised products + the reduction's NL products per column
template <class C> DVN_HD constexpr bool fe_mul_add_raw_fits() {
    const unsigned __int128 cols = (unsigned __int128)C::NL * (3ull << LB) * LMASK + 2 * (unsigned __int128)C::NL * LMASK * LMASK + ((unsigned __int128)1 << 36);
    return (cols >> 64) == 0;
}
// ... and with c raw as well (limbs below 2 * 2^29, as fe_neg_raw gives)
template <class C> DVN_HD constexpr bool fe_mul_add_raw2_fits() {
    const unsigned __int128 cols = (unsigned __int128)C::NL * (5ull << LB) * LMASK + (unsigned __int128)C::NL * LMASK * LMASK + ((unsigned __int128)1 << 36);
    return (cols >> 64) == 0;
}
// K p - b with no carry pass (the offset pattern of fe_sub_raw): limbs below 2^SH + 2^29
template <class C, int K, int SH> DVN_HD Fe<C> fe_neg_raw(const Fe<C>& b) {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const u32 k = fe_kp_limb<C, K>(i) + (i < C::NL - 1 ? (1u << SH) : 0u) - (i > 0 ? (1u << (SH - LB)) : 0u);
        r.v[i] = k - b.v[i];
    }
    return r;
}
// one unsigned carry pass: limbs < 2^32 in, limbs 0..NL-2 normalised out
template <class C> DVN_HD Fe<C> fe_normalise(const Fe<C>& a) {
    Fe<C> r = a;
#pragma unroll
    for (int i = 0; i < C::NL - 1; ++i) {
        r.v[i + 1] += r.v[i] >> LB;
        r.v[i] &= LMASK;
    }
    return r;
}
// (a*b + c*d) / 2^261 mod p with ONE Montgomery reduction.  Limbs of all four < 2^29 (top limb free), values loose.
template <class C> DVN_HD Fe<C> fe_mul_add(const Fe<C>& a, const Fe<C>& b, const Fe<C>& c, const Fe<C>& d) {
    constexpr int NL = C::NL;
    u64 T[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) T[k] = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i)
#pragma unroll
        for (int j = 0; j < NL; ++j) { T[i + j] += (u64)a.v[i] * b.v[j]; T[i + j] += (u64)c.v[i] * d.v[j]; }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const u32 m = ((u32)T[k] * C::inv) & LMASK;
#pragma unroll
        for (int j = 0; j < NL; ++j) T[k + j] += (u64)m * C::p(j);
        T[k + 1] += T[k] >> LB;
    }
    Fe<C> r;
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) { T[k + 1] += T[k] >> LB; r.v[k - NL] = (u32)T[k] & LMASK; }
    r.v[NL - 1] = (u32)T[2 * NL - 1];
    return r;
}
// loose -> weak form: multiply by 1 (internal form of 1 is 2^261 mod p)
template <class C> DVN_HD Fe<C> fe_reduce_loose(const Fe<C>& a) { return fe_mul<C>(a, fe_one<C>()); }
// x == 0 (mod p) for a loose x < 16p: x = k p with k = x_0 * p^-1 mod 2^29; k >= 16 settles it in three instructions
template <class C> DVN_NOINLINE bool fe_is_zero_loose_slow(Fe<C> a) { return fe_is_zero<C>(fe_reduce_loose<C>(a)); }
template <class C> DVN_HD bool fe_is_zero_loose(const Fe<C>& a) {
    const u32 k = (a.v[0] * C::pinv) & LMASK;
    if (__builtin_expect(k >= 16, 1)) return false;      // 1 - 2^-25 of the time
    return fe_is_zero_loose_slow<C>(a);                   // real call: keeps the 162-mad reduction out of the hot loop
}

// x == 0 (mod p) for weak-form x: x is 0 or p
template <class C> DVN_HD bool fe_is_zero(const Fe<C>& a) {
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) { z |= a.v[i]; e |= a.v[i] ^ C::p(i); }
    return z == 0 || e == 0;
}
template <class C> DVN_HD bool fe_eq(const Fe<C>& a, const Fe<C>& b) { return fe_is_zero<C>(fe_sub<C>(a, b)); }

// weak form -> canonical [0, p)
template <class C> DVN_HD Fe<C> fe_canonical(const Fe<C>& a) {
    i32 d[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) d[i] = (i32)a.v[i] - (i32)C::p(i);
    fe_carry<C>(d);
    const bool neg = d[C::NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = neg ? a.v[i] : (u32)d[i];
    return r;
}

// bits [OFF, OFF + 32) of a value with normalised limbs (get(i) = limb i); bits beyond 32 are dropped
template <class C, int OFF, class G> DVN_HD constexpr u32 fe_window32(G get) {
    constexpr int i0 = OFF / LB, o = OFF % LB;
    u32 w = get(i0) >> o;
    if (i0 + 1 < C::NL) w |= get(i0 + 1) << (LB - o);
    if (i0 + 2 < C::NL && 2 * LB - o < 32) w |= get(i0 + 2) << (2 * LB - o);
    return w;
}
// Any value below 2^(BITS + 6) with normalised limbs -> weak form (< 2p), exactly: subtract q p where q is the quotient
// estimated from the 32-bit window of the value that starts 22 bits below p's top bit.  With t that window, P = p >> OFF
// (2^21 <= P < 2^22) and M = floor(2^32 / (P + 1)):  t M / 2^32 <= t / (P + 1) <= x / p  and  x / p - t M / 2^32 < 2^-14 + 2^-4,
// so q = floor(t M / 2^32) is floor(x / p) or one less and x - q p lies in [0, 2p).  x - q p is formed without borrows as
// sum_i (x_i + q k_i) 2^(29 i) - q 2^(29 NL) with k_0 = 2^29 - p_0, k_i = 2^29 - p_i - 1: every term is non-negative, and the
// final -q 2^(29 NL) is the part of the top limb that masking removes.
template <class C> DVN_HD Fe<C> fe_reduce_full(const Fe<C>& x) {
    constexpr int NL = C::NL, OFF = C::BITS - 22;
    static_assert(C::BITS + 6 <= LB * NL, "headroom of the top limb");
    constexpr u32 P = fe_window32<C, OFF>([](int i) constexpr { return C::p(i); });
    static_assert(P >= (1u << 21) && P < (1u << 22), "window position");
    constexpr u32 M = (u32)((1ull << 32) / (P + 1));
    const u32 t = fe_window32<C, OFF>([&x](int i) { return x.v[i]; });
    const u32 q = (u32)(((u64)t * M) >> 32);
    u64 l[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = (u64)q * (u32)((1u << LB) - C::p(i) - (i > 0 ? 1u : 0u)) + x.v[i];
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) { l[i + 1] += l[i] >> LB; r.v[i] = (u32)l[i] & LMASK; }
    r.v[NL - 1] = (u32)l[NL - 1] & LMASK;
    return r;
}

// ---- external EW x u32 words (gnark's little-endian u64 limbs) <-> 29-bit limbs, pure bit slicing -------------------
template <class C> DVN_HD Fe<C> fe_unpack(const u32 (&w)[C::EW]) {
    Fe<C> r;
#pragma unroll
    for (int k = 0; k < C::NL; ++k) {
        const int bit = LB * k, wi = bit >> 5, sh = bit & 31;
        u64 two = (wi < C::EW) ? (u64)w[wi < C::EW ? wi : 0] : 0ull;
        if (wi + 1 < C::EW) two |= (u64)w[wi + 1 < C::EW ? wi + 1 : 0] << 32;
        r.v[k] = (u32)(two >> sh) & LMASK;
    }
    return r;
}
// a must be canonical or at least < 2^(32 EW) with limbs < 2^29
template <class C> DVN_HD void fe_pack(const Fe<C>& a, u32 (&w)[C::EW]) {
#pragma unroll
    for (int i = 0; i < C::EW; ++i) {
        const int bit = 32 * i, k = bit / LB, sh = bit - LB * k;   // word i starts inside limb k at offset sh
        u64 acc = (u64)a.v[k] >> sh;
        if (k + 1 < C::NL) acc |= (u64)a.v[k + 1 < C::NL ? k + 1 : 0] << (LB - sh);
        if (k + 2 < C::NL) acc |= (u64)a.v[k + 2 < C::NL ? k + 2 : 0] << (2 * LB - sh);
        w[i] = (u32)acc;
    }
}
#define DVN_CONST_FE(fn, table)                                  \
    template <class C> DVN_HD Fe<C> fn() { Fe<C> r;             \
        _Pragma("unroll")                                        \
        for (int i = 0; i < C::NL; ++i) r.v[i] = C::table(i);    \
        return r; }
DVN_CONST_FE(fe_const_conv_in, conv_in)
DVN_CONST_FE(fe_const_conv_out, conv_out)
DVN_CONST_FE(fe_const_r2, r2)
template <class C> DVN_HD Fe<C> fe_small(u32 k) { Fe<C> r = fe_zero<C>(); r.v[0] = k; return r; }   // plain integer k < 2^29

// gnark Montgomery word (x * 2^(32 EW)) -> internal Montgomery form of x (x * 2^(29 NL))
template <class C> DVN_HD Fe<C> fe_from_ext_mont(const u32 (&w)[C::EW]) { return fe_mul<C>(fe_unpack<C>(w), fe_const_conv_in<C>()); }
// internal form of x -> gnark Montgomery word
template <class C> DVN_HD void fe_to_ext_mont(const Fe<C>& a, u32 (&w)[C::EW]) { fe_pack<C>(fe_canonical<C>(fe_mul<C>(a, fe_const_conv_out<C>())), w); }
// gnark Montgomery word -> canonical integer x: x * 2^(32 EW) * 2^EXT_SHIFT / 2^(29 NL)
template <class C> DVN_HD Fe<C> fe_ext_mont_to_canonical(const u32 (&w)[C::EW]) { return fe_canonical<C>(fe_mul<C>(fe_unpack<C>(w), fe_small<C>(1u << C::EXT_SHIFT))); }
// canonical integer -> internal Montgomery form
template <class C> DVN_HD Fe<C> fe_from_canonical(const Fe<C>& x) { return fe_mul<C>(x, fe_const_r2<C>()); }

// a^e for a 64-bit exponent (internal form)
template <class C> DVN_HD Fe<C> fe_pow_u64(const Fe<C>& a, u64 e) {
    Fe<C> acc = fe_one<C>();
    for (int i = 63; i >= 0; --i) {
        acc = fe_sqr<C>(acc);
        if ((e >> i) & 1) acc = fe_mul<C>(acc, a);
    }
    return acc;
}
// a^(p-2) (Fermat inversion); 0 -> 0 as gnark's Inverse
template <class C> DVN_HD Fe<C> fe_inv(const Fe<C>& a) {
    // exponent p-2 as 29-bit limbs (borrow propagated: BLS12-377's moduli are 1 mod 2^46)
    u32 e[C::NL];
    u32 borrow = 2;
#pragma unroll
    for (int k = 0; k < C::NL; ++k) {
        const u32 pk = C::p(k);
        if (pk >= borrow) { e[k] = pk - borrow; borrow = 0; }
        else { e[k] = pk + (1u << LB) - borrow; borrow = 1; }
    }
    Fe<C> acc = fe_one<C>();
    for (int k = C::NL - 1; k >= 0; --k) {
        u32 limb = 0;
#pragma unroll
        for (int j = 0; j < C::NL; ++j) limb = (j == k) ? e[j] : limb;       // select, no runtime-indexed register array
        for (int b = LB - 1; b >= 0; --b) {
            acc = fe_sqr<C>(acc);
            if ((limb >> b) & 1) acc = fe_mul<C>(acc, a);
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------------
// Quadratic extension Fe2 = Fe[u]/(u^2 + BETA), element a0 + a1*u (gnark-crypto internal/fptower.E2):
// BN254 BETA = 1, BLS12-377 BETA = 5.
// ------------------------------------------------------------------------------------------------------
template <class C> struct Fe2 { Fe<C> a0, a1; };
template <class C> struct Beta;
template <> struct Beta<FpCfg> { static constexpr int value = 1; };
template <> struct Beta<Fp377Cfg> { static constexpr int value = 5; };
template <> struct Beta<Fp381Cfg> { static constexpr int value = 1; };

typedef Fe<FpCfg> Fp;
typedef Fe<FrCfg> Fr;
typedef Fe2<FpCfg> Fp2;
typedef Fe<Fp377Cfg> Fp377;
typedef Fe<Fr377Cfg> Fr377;
typedef Fe2<Fp377Cfg> Fp2_377;
typedef Fe<Fp761Cfg> Fp761;
typedef Fe<Fp381Cfg> Fp381;
typedef Fe<Fr381Cfg> Fr381;
typedef Fe2<Fp381Cfg> Fp2_381;

// Uniform "field ops" interface used by the curve templates
template <class F> struct Ops;
template <class C> static DVN_NOINLINE Fe<C> fp_mul_call(Fe<C> a, Fe<C> b);
template <class C> static DVN_NOINLINE Fe<C> fp_sqr_call(Fe<C> a);
template <class C> struct Ops<Fe<C>> {
    typedef Fe<C> F;
    typedef C Cfg;
    static DVN_HD F zero() { return fe_zero<C>(); }
    static DVN_HD F one() { return fe_one<C>(); }
    static DVN_HD F add(const F& a, const F& b) { return fe_add<C>(a, b); }
    static DVN_HD F sub(const F& a, const F& b) { return fe_sub<C>(a, b); }
    static DVN_HD F dbl(const F& a) { return fe_dbl<C>(a); }
    static DVN_HD F neg(const F& a) { return fe_neg<C>(a); }
    // wide fields (BW6-761: 27 limbs, 108 registers of column accumulators): the multiplication is a real call
    static constexpr bool CALL_MUL = C::NL > 16;
    static DVN_HD F mul(const F& a, const F& b) { if constexpr (CALL_MUL) return fp_mul_call<C>(a, b); else return fe_mul<C>(a, b); }
    static DVN_HD F sqr(const F& a) { if constexpr (CALL_MUL) return fp_sqr_call<C>(a); else return fe_sqr<C>(a); }
    static DVN_HD F inv(const F& a) { return fe_inv<C>(a); }
    static DVN_HD bool is_zero(const F& a) { return fe_is_zero<C>(a); }
    static DVN_HD bool is_literal_zero(const F& a) { u32 z = 0;
#pragma unroll
        for (int i = 0; i < C::NL; ++i) z |= a.v[i];
        return z == 0; }
    static constexpr int WORDS = C::EW;   // external u32 words per element
    static constexpr bool OUTLINE = false;
    // Curve::acc_add_mixed uses the lazily reduced formulas where the 64-bit column accumulators have the headroom:
    // 3 x NL products of < 2^58 (+ limbs up to 2^30) must stay below 2^64 -> NL <= 16
    static constexpr bool LAZY = C::NL <= 16 && C::SLACK >= 169;
    static DVN_HD F load_internal(const u32* w) { u32 t[C::EW];
#pragma unroll
        for (int i = 0; i < C::EW; ++i) t[i] = w[i];
        return fe_unpack<C>(t); }
    static DVN_HD void store_internal(const F& a, u32* w) { u32 t[C::EW]; fe_pack<C>(fe_canonical<C>(a), t);
#pragma unroll
        for (int i = 0; i < C::EW; ++i) w[i] = t[i]; }
    static DVN_HD F from_ext(const u32* w) { u32 t[C::EW];
#pragma unroll
        for (int i = 0; i < C::EW; ++i) t[i] = w[i];
        return fe_from_ext_mont<C>(t); }
    static DVN_HD void to_ext(const F& a, u32* w) { u32 t[C::EW]; fe_to_ext_mont<C>(a, t);
#pragma unroll
        for (int i = 0; i < C::EW; ++i) w[i] = t[i]; }
};
// Base-field multiplication as a real call: extension-field code stays compact (an inlined G2 kernel exceeds 10^5
// instructions and needs long-branch relaxation)
template <class C> static DVN_NOINLINE Fe<C> fp_mul_call(Fe<C> a, Fe<C> b) { return fe_mul<C>(a, b); }
template <class C> static DVN_NOINLINE Fe<C> fp_sqr_call(Fe<C> a) { return fe_sqr<C>(a); }
template <class C> struct OpsCall : Ops<Fe<C>> {
    static DVN_HD Fe<C> mul(const Fe<C>& a, const Fe<C>& b) { return fp_mul_call<C>(a, b); }
    static DVN_HD Fe<C> sqr(const Fe<C>& a) { return fp_sqr_call<C>(a); }
};
template <class C> struct Ops<Fe2<C>> {
    typedef Fe2<C> F;
    static constexpr bool CALL_MUL = true;     // the base-field products are real calls (OpsCall)
    typedef OpsCall<C> B;
    static constexpr int BETA = Beta<C>::value;
    // BETA * x for the small positive BETA (1 or 5): additions only
    static DVN_HD Fe<C> mul_beta(const Fe<C>& x) {
        if constexpr (BETA == 1) return x;
        else { Fe<C> x2 = B::dbl(x), x4 = B::dbl(x2); return B::add(x4, x); }
    }
    static DVN_HD F zero() { return F{B::zero(), B::zero()}; }
    static DVN_HD F one() { return F{B::one(), B::zero()}; }
    static DVN_HD F add(const F& a, const F& b) { return F{B::add(a.a0, b.a0), B::add(a.a1, b.a1)}; }
    static DVN_HD F sub(const F& a, const F& b) { return F{B::sub(a.a0, b.a0), B::sub(a.a1, b.a1)}; }
    static DVN_HD F dbl(const F& a) { return F{B::dbl(a.a0), B::dbl(a.a1)}; }
    static DVN_HD F neg(const F& a) { return F{B::neg(a.a0), B::neg(a.a1)}; }
    static DVN_HD F mul(const F& a, const F& b) {        // Karatsuba: 3 base-field multiplications
        Fe<C> t0 = B::mul(a.a0, b.a0), t1 = B::mul(a.a1, b.a1);
        Fe<C> m = B::mul(B::add(a.a0, a.a1), B::add(b.a0, b.a1));
        return F{B::sub(t0, mul_beta(t1)), B::sub(B::sub(m, t0), t1)};
    }
    static DVN_HD F sqr(const F& a) {
        if constexpr (BETA == 1) {                       // (a0+a1)(a0-a1), 2 a0 a1
            Fe<C> m = B::mul(a.a0, a.a1);
            return F{B::mul(B::add(a.a0, a.a1), B::sub(a.a0, a.a1)), B::dbl(m)};
        } else {                                         // a0^2 - BETA a1^2, 2 a0 a1
            Fe<C> m = B::mul(a.a0, a.a1);
            return F{B::sub(B::sqr(a.a0), mul_beta(B::sqr(a.a1))), B::dbl(m)};
        }
    }
    static DVN_HD F inv(const F& a) {                    // (a0 - a1 u) / (a0^2 + BETA a1^2)
        Fe<C> d = B::inv(B::add(B::sqr(a.a0), mul_beta(B::sqr(a.a1))));
        return F{B::mul(a.a0, d), B::neg(B::mul(a.a1, d))};
    }
    static DVN_HD bool is_zero(const F& a) { return B::is_zero(a.a0) && B::is_zero(a.a1); }
    static DVN_HD bool is_literal_zero(const F& a) { return B::is_literal_zero(a.a0) && B::is_literal_zero(a.a1); }
    static constexpr int WORDS = 2 * C::EW;
    static constexpr bool OUTLINE = false;
    static constexpr bool LAZY = false;
    static DVN_HD F load_internal(const u32* w) { return F{B::load_internal(w), B::load_internal(w + C::EW)}; }
    static DVN_HD void store_internal(const F& a, u32* w) { B::store_internal(a.a0, w); B::store_internal(a.a1, w + C::EW); }
    static DVN_HD F from_ext(const u32* w) { return F{B::from_ext(w), B::from_ext(w + C::EW)}; }
    static DVN_HD void to_ext(const F& a, u32* w) { B::to_ext(a.a0, w); B::to_ext(a.a1, w + C::EW); }
};

}  // namespace dvn
