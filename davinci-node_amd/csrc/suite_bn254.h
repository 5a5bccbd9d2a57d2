// The BN254 suite of the curve-generic Groth16 orchestration (groth16_impl.h): the statetransition curve.  Its MSMs and
// computeH are the ones compiled in dvn_msm_g1.hip, dvn_msm_g2.hip and dvn_ntt.hip (cross-TU calls: the kernels are not
// instantiated a second time in the translation units that include this header).
#pragma once
#include "groth16_impl.h"

struct SuiteBn254 {
    typedef FrCfg FrC;
    typedef G1 G1T;
    typedef G2 G2T;
    typedef dvn_host::HG1 H1;
    typedef dvn_host::HG2 H2;
    static constexpr int SUITE = 0;
    static const u32* r_words() { return FR_MOD_W; }
    static G1::A g1_gen() { return g1_generator(); }
    static G2::A g2_gen() { return g2_generator(); }
    static int32_t domain_create(dvn_ctx* ctx, u32 log_n, dvn_domain** out) { return dvn_ntt_domain(ctx, log_n, out); }
    static int32_t compute_h(dvn_ctx* ctx, const dvn_domain* dom, u32* a, u32* b, u32* c) { return dvn_internal_compute_h_dev(ctx, dom, a, b, c); }
    static int32_t msm_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) { return dvn_internal_msm_submit_g1(ctx, p, off, n, s, l); }
    static int32_t msm_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, MsmLane** l) { return dvn_internal_msm_submit_g2(ctx, p, off, n, s, l); }
    static int32_t msm_share_submit_g1(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return dvn_internal_msm_share_submit_g1(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_share_submit_g2(dvn_ctx* ctx, const dvn_points* p, size_t off, size_t n, const u32* s, u32 rank, u32 world, MsmLane** l) { return dvn_internal_msm_share_submit_g2(ctx, p, off, n, s, rank, world, l); }
    static int32_t msm_collect_g1(dvn_ctx* ctx, MsmLane* l, H1::X* out) { return dvn_internal_msm_collect_g1(ctx, l, out); }
    static int32_t msm_collect_g2(dvn_ctx* ctx, MsmLane* l, H2::X* out) { return dvn_internal_msm_collect_g2(ctx, l, out); }
};

