// Host build of the device arithmetic headers, for CPU-side tests (tests/test_host_arith.py): the same
// bn254_field.h / bn254_curve.h the HIP kernels compile are exercised here with g++ against the oracle.
// This library is a test seam only; it is never loaded by the product path.
#include <cstring>
#include "bn254_curve.h"
#include "host_fast.h"
using namespace dvn;

template <class C> static void bin(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) {
    u32 wa[8], wb[8], wr[8]; memcpy(wa, a, 32); memcpy(wb, b, 32);
    Fe<C> x = fe_from_ext_mont<C>(wa), y = fe_from_ext_mont<C>(wb), z;
    switch (op) {
        case 0: z = fe_mul<C>(x, y); break;
        case 1: z = fe_add<C>(x, y); break;
        case 2: z = fe_sub<C>(x, y); break;
        case 3: z = fe_sqr<C>(x); break;
        case 4: z = fe_inv<C>(x); break;
        case 5: z = fe_neg<C>(x); break;
        default: z = fe_dbl<C>(x); break;
    }
    fe_to_ext_mont<C>(z, wr); memcpy(r, wr, 32);
}
extern "C" {
void hc_fp_op(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) { bin<FpCfg>(a, b, r, op); }
void hc_fr_op(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) { bin<FrCfg>(a, b, r, op); }
void hc_fr_ext_to_canonical(const uint32_t* a, uint32_t* r) {
    u32 wa[8], wr[8]; memcpy(wa, a, 32);
    fe_pack<FrCfg>(fe_ext_mont_to_canonical<FrCfg>(wa), wr); memcpy(r, wr, 32);
}
// raw slicing round trip (no Montgomery conversion)
void hc_pack_roundtrip(const uint32_t* a, uint32_t* r) {
    u32 wa[8], wr[8]; memcpy(wa, a, 32); fe_pack<FpCfg>(fe_unpack<FpCfg>(wa), wr); memcpy(r, wr, 32);
}
int hc_fp_is_zero_of_diff(const uint32_t* a, const uint32_t* b) {
    u32 wa[8], wb[8]; memcpy(wa, a, 32); memcpy(wb, b, 32);
    return fe_eq<FpCfg>(fe_from_ext_mont<FpCfg>(wa), fe_from_ext_mont<FpCfg>(wb)) ? 1 : 0;
}
// out = sum_i k_i * P_i with the device group law (double-and-add per point), canonical 4xu64 scalars
void hc_g1_msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) {
    G1::X acc = G1::infinity();
    for (size_t i = 0; i < n; ++i) {
        G1::A p = G1::affine_from_ext(pts + 16 * i);
        G1::X t = G1::infinity();
        for (int b = 255; b >= 0; --b) {
            t = G1::dbl(t);
            if ((k[8 * i + (b >> 5)] >> (b & 31)) & 1) t = G1::add_mixed(t, p);
        }
        acc = G1::add(acc, t);
    }
    G1::affine_to_ext(G1::to_affine(acc), out);
}
void hc_g2_msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) {
    G2::X acc = G2::infinity();
    for (size_t i = 0; i < n; ++i) {
        G2::A p = G2::affine_from_ext(pts + 32 * i);
        G2::X t = G2::infinity();
        for (int b = 255; b >= 0; --b) {
            t = G2::dbl(t);
            if ((k[8 * i + (b >> 5)] >> (b & 31)) & 1) t = G2::add_mixed(t, p);
        }
        acc = G2::add(acc, t);
    }
    G2::affine_to_ext(G2::to_affine(acc), out);
}
// a + b on affine images, through add_mixed (exercises doubling / inverse / infinity branches)
void hc_g1_add(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    G1::affine_to_ext(G1::to_affine(G1::add_mixed(G1::from_affine(G1::affine_from_ext(a)), G1::affine_from_ext(b))), out);
}
void hc_g1_add_full(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    // push both through a non-trivial ZZ by doubling and adding back, then the full add
    G1::X pa = G1::from_affine(G1::affine_from_ext(a)), pb = G1::from_affine(G1::affine_from_ext(b));
    G1::X pa2 = G1::add(G1::dbl(pa), G1::neg(pa));   // = a with ZZ != 1 (if a finite)
    G1::X pb2 = G1::add(G1::dbl(pb), G1::neg(pb));
    G1::affine_to_ext(G1::to_affine(G1::add(pa2, pb2)), out);
}
// sum_i (+/-) P_i through the lazily reduced accumulator of k_accumulate (sign bit 31 of idx[i] = negate)
void hc_g1_acc_chain(const uint32_t* pts, const uint32_t* idx, size_t n, uint32_t* out) {
    G1::X acc = G1::infinity();
    for (size_t i = 0; i < n; ++i) {
        G1::A p = G1::affine_from_ext(pts + 16 * (size_t)(idx[i] & 0x7fffffffu));
        acc = G1::acc_add_mixed_signed(acc, p, (idx[i] & 0x80000000u) != 0);     // as k_accumulate does
    }
    G1::affine_to_ext(G1::to_affine(G1::acc_normalize(acc)), out);
}
void hc_g2_add(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    G2::affine_to_ext(G2::to_affine(G2::add_mixed(G2::from_affine(G2::affine_from_ext(a)), G2::affine_from_ext(b))), out);
}
}

// ---- BLS12-377 (14 x 29-bit limbs, external 6 x u64; Fe2 with u^2 = -5) --------------------------------------------
template <class C> static void bin12(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) {
    u32 wa[C::EW], wb[C::EW], wr[C::EW]; memcpy(wa, a, 4 * C::EW); memcpy(wb, b, 4 * C::EW);
    Fe<C> x = fe_from_ext_mont<C>(wa), y = fe_from_ext_mont<C>(wb), z;
    switch (op) {
        case 0: z = fe_mul<C>(x, y); break;
        case 1: z = fe_add<C>(x, y); break;
        case 2: z = fe_sub<C>(x, y); break;
        case 3: z = fe_sqr<C>(x); break;
        case 4: z = fe_inv<C>(x); break;
        case 5: z = fe_neg<C>(x); break;
        default: z = fe_dbl<C>(x); break;
    }
    fe_to_ext_mont<C>(z, wr); memcpy(r, wr, 4 * C::EW);
}
template <class Cv> static void msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) {
    typename Cv::X acc = Cv::infinity();
    for (size_t i = 0; i < n; ++i) {
        typename Cv::A p = Cv::affine_from_ext(pts + (size_t)Cv::AW * i);
        typename Cv::X t = Cv::infinity();
        for (int b = 255; b >= 0; --b) {
            t = Cv::dbl(t);
            if ((k[8 * i + (b >> 5)] >> (b & 31)) & 1) t = Cv::add_mixed(t, p);
        }
        acc = Cv::add(acc, t);
    }
    Cv::affine_to_ext(Cv::to_affine(acc), out);
}
extern "C" {
void hc377_fp_op(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) { bin12<Fp377Cfg>(a, b, r, op); }
void hc377_fr_op(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) { bin12<Fr377Cfg>(a, b, r, op); }
void hc377_fr_ext_to_canonical(const uint32_t* a, uint32_t* r) {
    u32 wa[8], wr[8]; memcpy(wa, a, 32);
    fe_pack<Fr377Cfg>(fe_ext_mont_to_canonical<Fr377Cfg>(wa), wr); memcpy(r, wr, 32);
}
void hc377_g1_msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) { msm_naive<G1_377>(pts, k, n, out); }
void hc377_g2_msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) { msm_naive<G2_377>(pts, k, n, out); }
// lazily reduced accumulator chain over G1-377 (the k_accumulate loop body)
void hc377_g1_acc_chain(const uint32_t* pts, const uint32_t* idx, size_t n, uint32_t* out) {
    G1_377::X acc = G1_377::infinity();
    for (size_t i = 0; i < n; ++i) {
        G1_377::A p = G1_377::affine_from_ext(pts + (size_t)G1_377::AW * (idx[i] & 0x7fffffffu));
        acc = G1_377::acc_add_mixed_signed(acc, p, (idx[i] & 0x80000000u) != 0);     // as k_accumulate does
    }
    G1_377::affine_to_ext(G1_377::to_affine(G1_377::acc_normalize(acc)), out);
}
}

// ---- BLS12-381 (same shapes as BLS12-377; u^2 = -1; 255-bit scalar field on 9 limbs with R'/r = 70) ----------------------
extern "C" {
void hc381_fp_op(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) { bin12<Fp381Cfg>(a, b, r, op); }
void hc381_fr_op(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) { bin12<Fr381Cfg>(a, b, r, op); }
void hc381_fr_ext_to_canonical(const uint32_t* a, uint32_t* r) {
    u32 wa[8], wr[8]; memcpy(wa, a, 32);
    fe_pack<Fr381Cfg>(fe_ext_mont_to_canonical<Fr381Cfg>(wa), wr); memcpy(r, wr, 32);
}
void hc381_g1_msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) { msm_naive<G1_381>(pts, k, n, out); }
void hc381_g2_msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) { msm_naive<G2_381>(pts, k, n, out); }
// lazily reduced accumulator chain over G1-381 (the k_accumulate loop body)
void hc381_g1_acc_chain(const uint32_t* pts, const uint32_t* idx, size_t n, uint32_t* out) {
    G1_381::X acc = G1_381::infinity();
    for (size_t i = 0; i < n; ++i) {
        G1_381::A p = G1_381::affine_from_ext(pts + (size_t)G1_381::AW * (idx[i] & 0x7fffffffu));
        acc = G1_381::acc_add_mixed_signed(acc, p, (idx[i] & 0x80000000u) != 0);     // as k_accumulate does
    }
    G1_381::affine_to_ext(G1_381::to_affine(G1_381::acc_normalize(acc)), out);
}
}

// ---- BW6-761 (27 x 29-bit limbs, external 12 x u64; scalars 377 bits = 12 words) ----------------------------------------
template <class Cv, int KW> static void msm_naive_wide(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) {
    typename Cv::X acc = Cv::infinity();
    for (size_t i = 0; i < n; ++i) {
        typename Cv::A p = Cv::affine_from_ext(pts + (size_t)Cv::AW * i);
        typename Cv::X t = Cv::infinity();
        for (int b = 32 * KW - 1; b >= 0; --b) {
            t = Cv::dbl(t);
            if ((k[KW * i + (b >> 5)] >> (b & 31)) & 1) t = Cv::add_mixed(t, p);
        }
        acc = Cv::add(acc, t);
    }
    Cv::affine_to_ext(Cv::to_affine(acc), out);
}
extern "C" {
void hc761_fp_op(const uint32_t* a, const uint32_t* b, uint32_t* r, int op) { bin12<Fp761Cfg>(a, b, r, op); }
// the scalar path of the MSM: gnark Montgomery image (R = 2^384) of a BLS12-377 Fp element -> canonical words
void hc761_scalar_ext_to_canonical(const uint32_t* a, uint32_t* r) {
    u32 wa[12], wr[12]; memcpy(wa, a, 48);
    fe_pack<Fp377Cfg>(fe_ext_mont_to_canonical<Fp377Cfg>(wa), wr); memcpy(r, wr, 48);
}
void hc761_msm_naive(const uint32_t* pts, const uint32_t* k, size_t n, uint32_t* out) { msm_naive_wide<G_761, 12>(pts, k, n, out); }
void hc761_acc_chain(const uint32_t* pts, const uint32_t* idx, size_t n, uint32_t* out) {
    G_761::X acc = G_761::infinity();
    for (size_t i = 0; i < n; ++i) {
        G_761::A p = G_761::affine_from_ext(pts + (size_t)G_761::AW * (idx[i] & 0x7fffffffu));
        acc = G_761::acc_add_mixed_signed(acc, p, (idx[i] & 0x80000000u) != 0);     // as k_accumulate does
    }
    G_761::affine_to_ext(G_761::to_affine(G_761::acc_normalize(acc)), out);
}
}


// ---- the 64-bit-limb host finishing arithmetic (host_fast.h OpsFqN / OpsFq2N) of the wider fields ------------------------------
// a, b, r: Montgomery images (R = 2^(64 NW)), NW uint64 each.  op: 0 mul, 1 add, 2 sub, 4 inv.  which: 0 BLS12-377 Fp, 1 BLS12-381 Fp,
// 2 BW6-761 Fp.
template <class O> static void hf_bin(const uint64_t* a, const uint64_t* b, uint64_t* r, int op) {
    typename O::F x, y, z;
    memcpy(x.l, a, sizeof(x.l)); memcpy(y.l, b, sizeof(y.l));
    switch (op) { case 0: z = O::mul(x, y); break; case 1: z = O::add(x, y); break; case 2: z = O::sub(x, y); break; default: z = O::inv(x); break; }
    memcpy(r, z.l, sizeof(z.l));
}
extern "C" void hc_hostfast_op(int which, const uint64_t* a, const uint64_t* b, uint64_t* r, int op) {
    if (which == 0) hf_bin<dvn_host::OpsFqN<Fp377Cfg, 6>>(a, b, r, op);
    else if (which == 1) hf_bin<dvn_host::OpsFqN<Fp381Cfg, 6>>(a, b, r, op);
    else hf_bin<dvn_host::OpsFqN<Fp761Cfg, 12>>(a, b, r, op);
}
// device internal image (x * 2^(29 NL), packed little-endian words) -> Montgomery image x * 2^(64 NW)
extern "C" void hc_hostfast_from_device(int which, const uint32_t* w, uint64_t* r) {
    if (which == 0) { auto v = dvn_host::OpsFqN<Fp377Cfg, 6>::from_device(w); memcpy(r, v.l, sizeof(v.l)); }
    else if (which == 1) { auto v = dvn_host::OpsFqN<Fp381Cfg, 6>::from_device(w); memcpy(r, v.l, sizeof(v.l)); }
    else { auto v = dvn_host::OpsFqN<Fp761Cfg, 12>::from_device(w); memcpy(r, v.l, sizeof(v.l)); }
}
// Fp2 product over BLS12-377 (u^2 = -5) / BLS12-381 (u^2 = -1): a, b, r = 12 uint64 (a0 | a1)
extern "C" void hc_hostfast_fp2_mul(int which, const uint64_t* a, const uint64_t* b, uint64_t* r, int inv) {
    if (which == 0) { typedef dvn_host::OpsFq2N<Fp377Cfg, 6, 5> O; O::F x, y; memcpy(&x, a, 96); memcpy(&y, b, 96); O::F z = inv ? O::inv(x) : O::mul(x, y); memcpy(r, &z, 96); }
    else { typedef dvn_host::OpsFq2N<Fp381Cfg, 6, 1> O; O::F x, y; memcpy(&x, a, 96); memcpy(&y, b, 96); O::F z = inv ? O::inv(x) : O::mul(x, y); memcpy(r, &z, 96); }
}

// ---- carry-free forms of the NTT butterflies, on raw limb arrays (NL uint32 each, no conversion) --------------------------
// field: 0 BN254 Fr, 1 BLS12-377 Fr, 2 BLS12-381 Fr, 3 BLS12-377 Fp (BW6-761's scalar field).
// op: 0 fe_reduce_full(a), 1 fe_sub_raw<8, 29>(a, b), 2 fe_sub_raw<16, 30>(a, b), 3 fe_sub_raw<2, 29>(a, b), 4 fe_normalise(a),
//     5 fe_mul(a, b) with a raw (limbs up to 2.5 * 2^30) and b normalised, 6 fe_canonical(fe_reduce_full(a))
template <class C> static void raw_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* r) {
    Fe<C> x, y, z;
    for (int i = 0; i < C::NL; ++i) { x.v[i] = a[i]; y.v[i] = b[i]; }
    switch (op) {
        case 0: z = fe_reduce_full<C>(x); break;
        case 1: z = fe_sub_raw<C, 8, LB>(x, y); break;
        case 2: z = fe_sub_raw<C, 16, 30>(x, y); break;
        case 3: z = fe_sub_raw<C, 2, LB>(x, y); break;
        case 4: z = fe_normalise<C>(x); break;
        case 5: z = fe_mul<C>(x, y); break;
        default: z = fe_canonical<C>(fe_reduce_full<C>(x)); break;
    }
    for (int i = 0; i < C::NL; ++i) r[i] = z.v[i];
}
extern "C" int hc_raw_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* r) {
    switch (field) {
        case 0: raw_op<FrCfg>(op, a, b, r); return FrCfg::NL;
        case 1: raw_op<Fr377Cfg>(op, a, b, r); return Fr377Cfg::NL;
        case 2: raw_op<Fr381Cfg>(op, a, b, r); return Fr381Cfg::NL;
        default: raw_op<Fp377Cfg>(op, a, b, r); return Fp377Cfg::NL;
    }
}

// ---- the one-bucket-set reduction, host half (host_fast.h bitsum_combine_pieces) ------------------------------------------------
// The pieces are formed here exactly as the device kernels define them -- row / column sums of the bucket matrix (k_marginal_level),
// subset sums by weight bit in spans of BITSUM_SPAN (k_marginal_bitsums: bit j of hi for the rows, bit j of lo + 1 for the columns)
// -- over B_b = [ks[b]] G on BN254 G1, then combined.  out: sum_b (b + 1) B_b as a G1Jac image (X | Y | Z, Montgomery words).
#include <vector>
extern "C" void hc_bitsum_reduction(uint32_t kh, uint32_t kl, const uint32_t* ks, uint64_t* out_jac) {
    typedef dvn_host::HG1 HG;
    typedef dvn_host::OpsFq O;
    const uint32_t RH = 1u << kh, RL = 1u << kl, SPAN = dvn_host::BITSUM_SPAN;
    const HG::X G{O::one(), O::add(O::one(), O::one()), O::one(), O::one()};                     // (1, 2)
    std::vector<HG::X> R(RH, HG::infinity()), Cs(RL, HG::infinity());
    for (uint32_t hi = 0; hi < RH; ++hi)
        for (uint32_t lo = 0; lo < RL; ++lo) {
            const HG::X B = HG::mul_scalar_n(G, &ks[(size_t)hi * RL + lo], 1);
            R[hi] = HG::add(R[hi], B); Cs[lo] = HG::add(Cs[lo], B);
        }
    const uint32_t sr = (RH + SPAN - 1) / SPAN, sc = (RL + SPAN - 1) / SPAN;
    std::vector<HG::X> P(dvn_host::bitsum_pieces(kh, kl), HG::infinity());
    for (uint32_t j = 0; j < kh; ++j)
        for (uint32_t s = 0; s < sr; ++s)
            for (uint32_t e = s * SPAN; e < RH && e < (s + 1) * SPAN; ++e)
                if ((e >> j) & 1) P[(size_t)j * sr + s] = HG::add(P[(size_t)j * sr + s], R[e]);
    for (uint32_t j = 0; j <= kl; ++j)
        for (uint32_t s = 0; s < sc; ++s)
            for (uint32_t e = s * SPAN; e < RL && e < (s + 1) * SPAN; ++e)
                if (((e + 1) >> j) & 1) P[(size_t)kh * sr + (size_t)j * sc + s] = HG::add(P[(size_t)kh * sr + (size_t)j * sc + s], Cs[e]);
    const HG::X r = dvn_host::bitsum_combine_pieces<HG>([&](size_t i) { return P[i]; }, kh, kl);
    HG::write_jac(r, out_jac);
}

// launch planning (host_fast.h)
extern "C" uint32_t hc_ntt_pass_plan(uint32_t log_n, uint32_t kmax, int dif, uint32_t* ks) { return dvn_host::ntt_pass_plan(log_n, kmax, dif != 0, ks); }
extern "C" void hc_expand_plan(uint64_t n, uint32_t scalar_bits, uint64_t point_bytes, uint32_t min_log, uint64_t max_bytes, uint32_t* c, uint32_t* nexp) {
    dvn_host::expand_plan_pure((size_t)n, scalar_bits, (size_t)point_bytes, min_log, (size_t)max_bytes, c, nexp);
}
