// BN254 Fp / Fr arithmetic for CDNA4 (gfx950), shared by device kernels and the host-side combine code.
//
// Replaces, on the device, gnark-crypto's ecc/bn254/fp and ecc/bn254/fr Element arithmetic that
// groth16.Prove (called at /root/reference/prover/prover_cpu.go:37, prover_gpu.go:33) spends its time in.
//
// Representation (MI355X-first, see DESIGN.md "Field arithmetic"):
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

constexpr int NL = 9;                 // limbs
constexpr int LB = 29;                // bits per limb
constexpr u32 LMASK = (1u << LB) - 1;

// Constant tables as constexpr functions of a compile-time index: after unrolling they become literals.
#define DVN_TABLE(name, ...)                                             \
    static DVN_HD constexpr u32 name(int i) {                            \
        constexpr u32 t[NL] = {__VA_ARGS__};                             \
        return t[i];                                                     \
    }

struct FpCfg {   // base field, p: /root/reference/config/statetransition_vkey.sol:42
    DVN_TABLE(p, 0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu)
    DVN_TABLE(p2, 0x10f9fa8eu, 0x0208c16du, 0x18e5469eu, 0x05aa45a1u, 0x0b0bb2f0u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu)
    DVN_TABLE(one, 0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u)
    DVN_TABLE(r2, 0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u, 0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u)
    DVN_TABLE(conv_in, 0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u)
    DVN_TABLE(conv_out, 0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u)
    DVN_TABLE(p4, 0x01f3f51cu, 0x041182dbu, 0x11ca8d3cu, 0x0b548b43u, 0x161765e0u, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u)
    DVN_TABLE(p8, 0x03e7ea38u, 0x082305b6u, 0x03951a78u, 0x16a91687u, 0x0c2ecbc0u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u)
    static constexpr u32 inv = 0x04866389u;   // -p^-1 mod 2^29
    static constexpr u32 pinv = 0x1b799c77u;  //  p^-1 mod 2^29
};
struct FrCfg {   // scalar field, r: /root/reference/config/statetransition_vkey.sol:43
    DVN_TABLE(p, 0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu)
    DVN_TABLE(p2, 0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u, 0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu)
    DVN_TABLE(one, 0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u)
    DVN_TABLE(r2, 0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au)
    DVN_TABLE(conv_in, 0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u)
    DVN_TABLE(conv_out, 0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u)
    DVN_TABLE(p4, 0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u)
    DVN_TABLE(p8, 0x00000008u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u)
    static constexpr u32 inv = 0x0fffffffu;   // -r^-1 mod 2^29
    static constexpr u32 pinv = 0x10000001u;  //  r^-1 mod 2^29
};

template <class C>
struct Fe {
    u32 v[NL];
};
typedef Fe<FpCfg> Fp;
typedef Fe<FrCfg> Fr;

template <class C> DVN_HD Fe<C> fe_zero() { Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = 0;
    return r; }
template <class C> DVN_HD Fe<C> fe_one() { Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = C::one(i);
    return r; }

// Signed carry propagation: limbs may be any int32; result limbs 0..7 in [0, 2^29), limb 8 keeps the sign.
template <class C> DVN_HD void fe_carry(i32 (&t)[NL]) {
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
        i32 c = t[i] >> LB;            // arithmetic shift: floor division
        t[i] &= (i32)LMASK;
        t[i + 1] += c;
    }
}

// r = a + b, weak form in / weak form out
template <class C> DVN_HD Fe<C> fe_add(const Fe<C>& a, const Fe<C>& b) {
    i32 s[NL], d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { s[i] = (i32)(a.v[i] + b.v[i]); d[i] = s[i] - (i32)C::p2(i); }
    fe_carry<C>(s); fe_carry<C>(d);
    const bool neg = d[NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = (u32)(neg ? s[i] : d[i]);
    return r;
}
// r = a - b
template <class C> DVN_HD Fe<C> fe_sub(const Fe<C>& a, const Fe<C>& b) {
    i32 s[NL], d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { d[i] = (i32)a.v[i] - (i32)b.v[i]; s[i] = d[i] + (i32)C::p2(i); }
    fe_carry<C>(s); fe_carry<C>(d);
    const bool neg = d[NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = (u32)(neg ? s[i] : d[i]);
    return r;
}
template <class C> DVN_HD Fe<C> fe_dbl(const Fe<C>& a) { return fe_add<C>(a, a); }
template <class C> DVN_HD Fe<C> fe_neg(const Fe<C>& a) { return fe_sub<C>(fe_zero<C>(), a); }

// Montgomery product a*b/2^261 mod p.  Inputs: limbs < 2^30, values < 13p.  Output weak form (< 1.03p .. 2p).
template <class C> DVN_HD Fe<C> fe_mul(const Fe<C>& a, const Fe<C>& b) {
    u64 T[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) T[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
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
    u64 T[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) T[k] = 0;
    u32 a2[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) a2[i] = a.v[i] << 1;          // < 2^31 for limbs < 2^30
#pragma unroll
    for (int i = 0; i < NL; ++i) {
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
    i32 d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (i32)a.v[i] + (i32)fe_kp<C, K>(i) - (i32)b.v[i];
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// K p - b
template <class C, int K> DVN_HD Fe<C> fe_neg_lazy(const Fe<C>& b) {
    i32 d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (i32)fe_kp<C, K>(i) - (i32)b.v[i];
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// a + b with one carry pass, no reduction: value a + b, limbs normalised
template <class C> DVN_HD Fe<C> fe_add_lazy(const Fe<C>& a, const Fe<C>& b) {
    i32 d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (i32)(a.v[i] + b.v[i]);
    fe_carry<C>(d);
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = (u32)d[i];
    return r;
}
// x >= K p ? x - K p : x   (x loose with normalised limbs)
template <class C, int K> DVN_HD Fe<C> fe_csub(const Fe<C>& x) {
    i32 d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (i32)x.v[i] - (i32)fe_kp<C, K>(i);
    fe_carry<C>(d);
    const bool neg = d[NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = neg ? x.v[i] : (u32)d[i];
    return r;
}
// a + b without carry propagation: limbs < 2^30; only valid as the subtrahend of fe_sub_lazy or a fe_mul operand
template <class C> DVN_HD Fe<C> fe_add_nocarry(const Fe<C>& a, const Fe<C>& b) {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// (a*b + c*d) / 2^261 mod p with ONE Montgomery reduction.  Limbs of all four < 2^29 (top limb free), values loose.
template <class C> DVN_HD Fe<C> fe_mul_add(const Fe<C>& a, const Fe<C>& b, const Fe<C>& c, const Fe<C>& d) {
    u64 T[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; ++k) T[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
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
    for (int i = 0; i < NL; ++i) { z |= a.v[i]; e |= a.v[i] ^ C::p(i); }
    return z == 0 || e == 0;
}
template <class C> DVN_HD bool fe_eq(const Fe<C>& a, const Fe<C>& b) { return fe_is_zero<C>(fe_sub<C>(a, b)); }

// weak form -> canonical [0, p)
template <class C> DVN_HD Fe<C> fe_canonical(const Fe<C>& a) {
    i32 d[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d[i] = (i32)a.v[i] - (i32)C::p(i);
    fe_carry<C>(d);
    const bool neg = d[NL - 1] < 0;
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = neg ? a.v[i] : (u32)d[i];
    return r;
}

// ---- external 8 x u32 (= 4 x u64 little-endian) <-> 29-bit limbs, pure bit slicing -------------------
template <class C> DVN_HD Fe<C> fe_unpack(const u32 (&w)[8]) {
    Fe<C> r;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int bit = LB * k, wi = bit >> 5, sh = bit & 31;
        u64 two = (u64)w[wi] | ((wi + 1 < 8) ? ((u64)w[wi + 1] << 32) : 0ull);
        r.v[k] = (u32)(two >> sh) & LMASK;
    }
    return r;
}
// a must be canonical or at least < 2^256 with limbs < 2^29
template <class C> DVN_HD void fe_pack(const Fe<C>& a, u32 (&w)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int bit = 32 * i, k = bit / LB, sh = bit - LB * k;   // word i starts inside limb k at offset sh
        u64 acc = (u64)a.v[k] >> sh;
        if (k + 1 < NL) acc |= (u64)a.v[k + 1] << (LB - sh);
        if (k + 2 < NL) acc |= (u64)a.v[k + 2] << (2 * LB - sh);
        w[i] = (u32)acc;
    }
}
template <class C> DVN_HD Fe<C> fe_const_conv_in() { Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = C::conv_in(i);
    return r; }
template <class C> DVN_HD Fe<C> fe_const_conv_out() { Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = C::conv_out(i);
    return r; }
template <class C> DVN_HD Fe<C> fe_const_r2() { Fe<C> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.v[i] = C::r2(i);
    return r; }
template <class C> DVN_HD Fe<C> fe_small(u32 k) { Fe<C> r = fe_zero<C>(); r.v[0] = k; return r; }   // plain integer k < 2^29

// gnark Montgomery word (x * 2^256) -> internal Montgomery form of x (x * 2^261)
template <class C> DVN_HD Fe<C> fe_from_ext_mont(const u32 (&w)[8]) { return fe_mul<C>(fe_unpack<C>(w), fe_const_conv_in<C>()); }
// internal form of x -> gnark Montgomery word
template <class C> DVN_HD void fe_to_ext_mont(const Fe<C>& a, u32 (&w)[8]) { fe_pack<C>(fe_canonical<C>(fe_mul<C>(a, fe_const_conv_out<C>())), w); }
// gnark Montgomery word -> canonical integer x (as packed words): x*2^256 * 2^5 / 2^261
template <class C> DVN_HD Fe<C> fe_ext_mont_to_canonical(const u32 (&w)[8]) { return fe_canonical<C>(fe_mul<C>(fe_unpack<C>(w), fe_small<C>(32))); }
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
    // exponent p-2 as 29-bit limbs: p(0) is odd and >= 3 for both fields, so only limb 0 changes
    Fe<C> acc = fe_one<C>();
    for (int k = NL - 1; k >= 0; --k) {
        u32 limb = 0;
        switch (k) {   // runtime k, compile-time table
            case 0: limb = C::p(0) - 2; break; case 1: limb = C::p(1); break; case 2: limb = C::p(2); break;
            case 3: limb = C::p(3); break; case 4: limb = C::p(4); break; case 5: limb = C::p(5); break;
            case 6: limb = C::p(6); break; case 7: limb = C::p(7); break; default: limb = C::p(8); break;
        }
        for (int b = LB - 1; b >= 0; --b) {
            acc = fe_sqr<C>(acc);
            if ((limb >> b) & 1) acc = fe_mul<C>(acc, a);
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2+1), element a0 + a1*u  (gnark-crypto ecc/bn254/internal/fptower.E2)
// ------------------------------------------------------------------------------------------------------
struct Fp2 {
    Fp a0, a1;
};

// Uniform "field ops" interface used by the curve templates: Ops<Fp>, Ops<Fp2>
template <class F> struct Ops;
template <> struct Ops<Fp> {
    static DVN_HD Fp zero() { return fe_zero<FpCfg>(); }
    static DVN_HD Fp one() { return fe_one<FpCfg>(); }
    static DVN_HD Fp add(const Fp& a, const Fp& b) { return fe_add<FpCfg>(a, b); }
    static DVN_HD Fp sub(const Fp& a, const Fp& b) { return fe_sub<FpCfg>(a, b); }
    static DVN_HD Fp dbl(const Fp& a) { return fe_dbl<FpCfg>(a); }
    static DVN_HD Fp neg(const Fp& a) { return fe_neg<FpCfg>(a); }
    static DVN_HD Fp mul(const Fp& a, const Fp& b) { return fe_mul<FpCfg>(a, b); }
    static DVN_HD Fp sqr(const Fp& a) { return fe_sqr<FpCfg>(a); }
    static DVN_HD Fp inv(const Fp& a) { return fe_inv<FpCfg>(a); }
    static DVN_HD bool is_zero(const Fp& a) { return fe_is_zero<FpCfg>(a); }
    static DVN_HD bool is_literal_zero(const Fp& a) { u32 z = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) z |= a.v[i];
        return z == 0; }
    static constexpr int WORDS = 8;   // external u32 words per element
    static constexpr bool OUTLINE = false;
    static constexpr bool LAZY = true;    // Curve::acc_add_mixed uses the lazily reduced formulas
    typedef FpCfg Cfg;
    static DVN_HD Fp load_internal(const u32* w) { u32 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = w[i];
        return fe_unpack<FpCfg>(t); }
    static DVN_HD void store_internal(const Fp& a, u32* w) { u32 t[8]; fe_pack<FpCfg>(fe_canonical<FpCfg>(a), t);
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = t[i]; }
    static DVN_HD Fp from_ext(const u32* w) { u32 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = w[i];
        return fe_from_ext_mont<FpCfg>(t); }
    static DVN_HD void to_ext(const Fp& a, u32* w) { u32 t[8]; fe_to_ext_mont<FpCfg>(a, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = t[i]; }
};
// Fp multiplication as a real call: Fp2 code stays compact (an inlined G2 kernel exceeds 10^5 instructions)
static DVN_NOINLINE Fp fp_mul_call(Fp a, Fp b) { return fe_mul<FpCfg>(a, b); }
static DVN_NOINLINE Fp fp_sqr_call(Fp a) { return fe_sqr<FpCfg>(a); }
struct OpsFpCall : Ops<Fp> {
    static DVN_HD Fp mul(const Fp& a, const Fp& b) { return fp_mul_call(a, b); }
    static DVN_HD Fp sqr(const Fp& a) { return fp_sqr_call(a); }
};
template <> struct Ops<Fp2> {
    typedef OpsFpCall B;
    static DVN_HD Fp2 zero() { return Fp2{B::zero(), B::zero()}; }
    static DVN_HD Fp2 one() { return Fp2{B::one(), B::zero()}; }
    static DVN_HD Fp2 add(const Fp2& a, const Fp2& b) { return Fp2{B::add(a.a0, b.a0), B::add(a.a1, b.a1)}; }
    static DVN_HD Fp2 sub(const Fp2& a, const Fp2& b) { return Fp2{B::sub(a.a0, b.a0), B::sub(a.a1, b.a1)}; }
    static DVN_HD Fp2 dbl(const Fp2& a) { return Fp2{B::dbl(a.a0), B::dbl(a.a1)}; }
    static DVN_HD Fp2 neg(const Fp2& a) { return Fp2{B::neg(a.a0), B::neg(a.a1)}; }
    static DVN_HD Fp2 mul(const Fp2& a, const Fp2& b) {        // Karatsuba: 3 Fp multiplications
        Fp t0 = B::mul(a.a0, b.a0), t1 = B::mul(a.a1, b.a1);
        Fp m = B::mul(B::add(a.a0, a.a1), B::add(b.a0, b.a1));
        return Fp2{B::sub(t0, t1), B::sub(B::sub(m, t0), t1)};
    }
    static DVN_HD Fp2 sqr(const Fp2& a) {                       // (a0+a1)(a0-a1), 2 a0 a1
        Fp m = B::mul(a.a0, a.a1);
        return Fp2{B::mul(B::add(a.a0, a.a1), B::sub(a.a0, a.a1)), B::dbl(m)};
    }
    static DVN_HD Fp2 inv(const Fp2& a) {
        Fp d = B::inv(B::add(B::sqr(a.a0), B::sqr(a.a1)));
        return Fp2{B::mul(a.a0, d), B::neg(B::mul(a.a1, d))};
    }
    static DVN_HD bool is_zero(const Fp2& a) { return B::is_zero(a.a0) && B::is_zero(a.a1); }
    static DVN_HD bool is_literal_zero(const Fp2& a) { return B::is_literal_zero(a.a0) && B::is_literal_zero(a.a1); }
    static constexpr int WORDS = 16;
    static constexpr bool OUTLINE = false;
    static constexpr bool LAZY = false;
    static DVN_HD Fp2 load_internal(const u32* w) { return Fp2{B::load_internal(w), B::load_internal(w + 8)}; }
    static DVN_HD void store_internal(const Fp2& a, u32* w) { B::store_internal(a.a0, w); B::store_internal(a.a1, w + 8); }
    static DVN_HD Fp2 from_ext(const u32* w) { return Fp2{B::from_ext(w), B::from_ext(w + 8)}; }
    static DVN_HD void to_ext(const Fp2& a, u32* w) { B::to_ext(a.a0, w); B::to_ext(a.a1, w + 8); }
};

}  // namespace dvn
