// G2 instantiation of the MSM pipeline.
#include "msm_driver.h"

extern "C" int32_t dvn_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) try { return points_upload<G2>(ctx, xy, n, 2, out); } DVN_CATCH

extern "C" int32_t dvn_points_synthetic_g2(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out) try { return points_synthetic<G2>(ctx, seed, n, 2, g2_generator(), out); } DVN_CATCH

extern "C" int32_t dvn_msm_g2_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, int32_t* ticket) try {
    return msm_submit_api<G2>(ctx, pts, 2, off, n, scalars, flags, ticket);
} DVN_CATCH
extern "C" int32_t dvn_msm_g2_collect(dvn_ctx* ctx, int32_t ticket, uint64_t out_jac[24]) try { return msm_collect_api<G2>(ctx, 2, ticket, out_jac); } DVN_CATCH
extern "C" int32_t dvn_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[24]) try {
    return msm_full<G2>(ctx, pts, 2, off, n, scalars, flags, out_jac);
} DVN_CATCH

int32_t dvn_internal_points_download_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) {
    return points_download<G2>(ctx, pts, off, n, xy);
}

int32_t dvn_internal_msm_g2_dev(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, dvn_host::HG2::X* out) {
    if (pts->curve != 2) return fail(DVN_ERR_INVALID_ARG, "point set is not G2");
    return msm_dev_xyzz<G2>(ctx, pts, off, n, d_scalars, out);
}

int32_t dvn_internal_points_set_expanded_g2(dvn_ctx* ctx, dvn_points* pts, u32 c) { return points_set_expanded<G2>(ctx, pts, c); }

int32_t dvn_internal_msm_submit_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, MsmLane** lane) {
    if (pts->curve != 2) return fail(DVN_ERR_INVALID_ARG, "point set is not G2");
    return msm_submit<G2>(ctx, pts, off, n, (const uint64_t*)d_scalars, DVN_MSM_SCALARS_ON_DEVICE, lane);
}
int32_t dvn_internal_msm_collect_g2(dvn_ctx* ctx, MsmLane* lane, dvn_host::HG2::X* out) { return msm_collect<G2>(ctx, *lane, out); }
int32_t dvn_internal_msm_share_submit_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, u32 rank, u32 world, MsmLane** lane) {
    if (pts->curve != 2) return fail(DVN_ERR_INVALID_ARG, "point set is not G2");
    return msm_share_submit<G2>(ctx, pts, off, n, d_scalars, rank, world, lane);
}
