// Fp2 arithmetic with ONE COMPONENT PER LANE: the even lane of a pair holds a0, the odd lane a1 of
// a0 + a1 u.  A G2 bucket accumulator then costs a lane what a G1 accumulator costs (4 x NL limbs of state instead of
// 8 x NL), so the G2 hot loop runs fully inlined at the G1 kernel's occupancy instead of call-based at one wave per SIMD.
// The partner's limbs arrive by DPP quad_perm [1,0,3,2] (one v_mov_dpp per limb, no LDS); both lanes execute the same
// instruction stream:  mul = U*V + S*T with one Montgomery reduction (fe_mul_add), operands selected per lane,
//                      sqr = U*V (BETA = 1) with U, V selected per lane.
// Memory images are the ordinary Fp2 ones (a0 words, then a1 words): lane parity picks its half.
#pragma once
#include "bn254_field.h"
#include "bn254_curve.h"

namespace dvn {

template <class C> struct Fe2P { Fe<C> c; };

__device__ __forceinline__ u32 pair_odd() { return threadIdx.x & 1u; }
__device__ __forceinline__ u32 dpp_xor1(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }   // quad_perm [1,0,3,2]
template <class C> __device__ __forceinline__ Fe<C> fe_partner(const Fe<C>& a) {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = dpp_xor1(a.v[i]);
    return r;
}
template <class C> __device__ __forceinline__ Fe<C> fe_select(bool c, const Fe<C>& a, const Fe<C>& b) {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

template <class C> struct Ops<Fe2P<C>> {
    typedef Fe2P<C> F;
    typedef C Cfg;
    static constexpr int BETA = Beta<C>::value;
    static constexpr bool CALL_MUL = false, OUTLINE = false, LAZY = false;
    static constexpr int WORDS = 2 * C::EW;
    static __device__ __forceinline__ F zero() { return F{fe_zero<C>()}; }
    static __device__ __forceinline__ F one() { return F{fe_select<C>(pair_odd(), fe_zero<C>(), fe_one<C>())}; }
    static __device__ __forceinline__ F add(const F& a, const F& b) { return F{fe_add<C>(a.c, b.c)}; }
    static __device__ __forceinline__ F sub(const F& a, const F& b) { return F{fe_sub<C>(a.c, b.c)}; }
    static __device__ __forceinline__ F dbl(const F& a) { return F{fe_dbl<C>(a.c)}; }
    static __device__ __forceinline__ F neg(const F& a) { return F{fe_neg<C>(a.c)}; }
    // BETA * x, loose (x weak): additions with one carry pass each
    static __device__ __forceinline__ Fe<C> mul_beta_lazy(const Fe<C>& x) {
        if constexpr (BETA == 1) return x;
        else { const Fe<C> x2 = fe_add_lazy<C>(x, x), x4 = fe_add_lazy<C>(x2, x2); return fe_add_lazy<C>(x4, x); }   // BETA = 5
    }
    // even lane: a0 b0 + (4p - a1)(BETA b1);  odd lane: a0 b1 + a1 b0   (mine = own component, part = partner's)
    static __device__ __forceinline__ F mul(const F& a, const F& b) {
        const bool odd = pair_odd();
        const Fe<C> pa = fe_partner<C>(a.c), pb = fe_partner<C>(b.c);
        const Fe<C> U = fe_select<C>(odd, pa, a.c);                                  // a0
        // a1 | 4p - a1 with no carry pass: as the first operand of the second product its limbs may reach 2^30 (the columns of
        // fe_mul_add hold NL * (1 + 2 + 1) * 2^58 < 2^64 for NL <= 16)
        static_assert(C::NL <= 16, "column headroom of the raw operand");
        const Fe<C> S = fe_select<C>(odd, a.c, fe_neg_raw<C, 4, LB>(pa));
        const Fe<C> T = fe_select<C>(odd, pb, mul_beta_lazy(pb));                    // b0 | BETA b1
        return F{fe_mul_add<C>(U, b.c, S, T)};                                       // V = own b component
    }
    static __device__ __forceinline__ F sqr(const F& a) {
        if constexpr (BETA == 1) {        // even: (a0 + a1)(a0 + 2p - a1);  odd: (a0 + a0) a1
            const bool odd = pair_odd();
            const Fe<C> pa = fe_partner<C>(a.c);
            const Fe<C> U = fe_add_lazy<C>(pa, fe_select<C>(odd, pa, a.c));
            const Fe<C> V = fe_select<C>(odd, a.c, fe_sub_lazy<C, 2>(a.c, pa));
            return F{fe_mul<C>(U, V)};
        } else return mul(a, a);
    }
    static __device__ __forceinline__ bool is_zero(const F& a) {
        const u32 z = fe_is_zero<C>(a.c) ? 1u : 0u;
        return (z & dpp_xor1(z)) != 0;
    }
    static __device__ __forceinline__ bool is_literal_zero(const F& a) {
        u32 z = 0;
#pragma unroll
        for (int i = 0; i < C::NL; ++i) z |= a.c.v[i];
        z |= dpp_xor1(z);
        return z == 0;
    }
    // this lane's half of the Fp2 image at w (2 * EW words)
    static __device__ __forceinline__ F load_internal_half(const u32* __restrict__ w) {
        const u32* p = w + pair_odd() * C::EW;
        u32 t[C::EW];
#pragma unroll
        for (int i = 0; i < C::EW; i += 4) { const uint4 v = *reinterpret_cast<const uint4*>(p + i); t[i] = v.x; t[i + 1] = v.y; t[i + 2] = v.z; t[i + 3] = v.w; }
        return F{fe_unpack<C>(t)};
    }
    static __device__ __forceinline__ void store_internal_half(const F& a, u32* __restrict__ w) {     // a weak form
        u32* p = w + pair_odd() * C::EW;
        u32 t[C::EW];
        fe_pack<C>(fe_canonical<C>(a.c), t);
#pragma unroll
        for (int i = 0; i < C::EW; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(t[i], t[i + 1], t[i + 2], t[i + 3]);
    }
};

// paired point loads / stores: each lane moves its own halves of the ordinary Fp2 images
template <class Cv> __device__ __forceinline__ typename Cv::A ld_affine_paired(const u32* __restrict__ p) {
    typedef typename Cv::O O;
    return typename Cv::A{O::load_internal_half(p), O::load_internal_half(p + O::WORDS)};
}
template <class Cv> __device__ __forceinline__ typename Cv::X ld_xyzz_paired(const u32* __restrict__ p) {
    typedef typename Cv::O O;
    return typename Cv::X{O::load_internal_half(p), O::load_internal_half(p + O::WORDS), O::load_internal_half(p + 2 * O::WORDS),
                          O::load_internal_half(p + 3 * O::WORDS)};
}
template <class Cv> __device__ __forceinline__ void st_xyzz_paired(const typename Cv::X& pt, u32* __restrict__ p) {
    typedef typename Cv::O O;
    if (Cv::is_inf(pt)) {                       // canonical image of infinity: all zero
        const typename Cv::O::F z = O::zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) O::store_internal_half(z, p + j * O::WORDS);
        return;
    }
    O::store_internal_half(pt.x, p); O::store_internal_half(pt.y, p + O::WORDS);
    O::store_internal_half(pt.zz, p + 2 * O::WORDS); O::store_internal_half(pt.zzz, p + 3 * O::WORDS);
}

}  // namespace dvn
