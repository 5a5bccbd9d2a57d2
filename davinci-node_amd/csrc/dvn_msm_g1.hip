// G1 instantiation of the MSM pipeline + curve-independent point-set entry points.
#include <thread>
#include "msm_driver.h"

int32_t dvn_internal_points_download_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy);   // dvn_msm_g2.hip

extern "C" int32_t dvn_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy, size_t n, dvn_points** out) try { return points_upload<G1>(ctx, xy, n, 1, out); } DVN_CATCH

extern "C" int32_t dvn_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out) try { return points_synthetic<G1>(ctx, seed, n, 1, g1_generator(), out); } DVN_CATCH

extern "C" int32_t dvn_msm_g1_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, int32_t* ticket) try {
    return msm_submit_api<G1>(ctx, pts, 1, off, n, scalars, flags, ticket);
} DVN_CATCH
extern "C" int32_t dvn_msm_g1_collect(dvn_ctx* ctx, int32_t ticket, uint64_t out_jac[12]) try { return msm_collect_api<G1>(ctx, 1, ticket, out_jac); } DVN_CATCH
extern "C" int32_t dvn_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t out_jac[12]) try {
    return msm_full<G1>(ctx, pts, 1, off, n, scalars, flags, out_jac);
} DVN_CATCH

// ---- the entry point that allocates nothing ----------------------------------------------------------------------------
// dvn_msm_g1 keeps its sort / bucket / partial-sum arrays in a grow-only hipMalloc'ed workspace.  Callers that cannot allocate
// (tight serving loops, harnesses that forbid hipMalloc, code that must not disturb the allocator) get the same MSM over a FIXED scratch
// buffer: a file-scope device array of DVN_MSM_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the
// process ends.  Its device address differs per GPU: it is resolved once per context (= per device) at dvn_init, with that
// device current, and only that pointer -- never the array's name -- is handed to kernels and fills.
static constexpr size_t DVN_MSM_STATIC_SCRATCH_BYTES = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[DVN_MSM_STATIC_SCRATCH_BYTES];

int32_t dvn_internal_static_scratch_init(dvn_ctx* ctx) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(DVN_ERR_DEVICE, "hipGetSymbolAddress returned a NULL scratch address on device %d", ctx->device);
    ctx->static_ws = p; ctx->static_ws_bytes = DVN_MSM_STATIC_SCRATCH_BYTES;
    return DVN_OK;
}
// Same result as dvn_msm_g1 (same kernels), n <= DVN_MSM_STATIC_MAX_N, scalars already in device memory.  No hipMalloc / hipFree:
// scratch comes from the static array, clears are fill kernels, the result leaves through the lane's pinned buffer of dvn_init.
// All calls on one GPU share that GPU's array (it keeps the previous call's contents; everything that must start at zero is
// cleared per call).  Calls through this library are serialised per device -- a process-wide mutex per ordinal, held from
// submit to collect, because dvn_init may put several contexts on one GPU -- so they cannot corrupt each other; anything ELSE
// that touches the array at the same time (another process image of the library on the same GPU has its own copy) would race.
// The call synchronises its stream and finishes on the host: it is not capturable into a HIP graph.
static std::mutex static_scratch_mu[64];
extern "C" int32_t dvn_msm_g1_static(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* d_scalars_mont,
                                     uint32_t flags, uint64_t out_jac[12]) try {
    TRY(enter(ctx));
    if (!pts || !out_jac || (!d_scalars_mont && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->curve != 1 || pts->suite != 0) return fail(DVN_ERR_INVALID_ARG, "point set is not BN254 G1");
    if (!(flags & DVN_MSM_SCALARS_ON_DEVICE)) return fail(DVN_ERR_INVALID_ARG, "dvn_msm_g1_static takes device scalars only (DVN_MSM_SCALARS_ON_DEVICE): a host copy could allocate staging memory");
    if (n > DVN_MSM_STATIC_MAX_N) return fail(DVN_ERR_INVALID_ARG, "n = %zu exceeds DVN_MSM_STATIC_MAX_N = %u, the most the %zu-byte static scratch buffer serves", n, (unsigned)DVN_MSM_STATIC_MAX_N, DVN_MSM_STATIC_SCRATCH_BYTES);
    typedef FastHost<G1>::G HG;
    if (ctx->device < 0 || ctx->device >= 64) return fail(DVN_ERR_INVALID_ARG, "device ordinal %d out of range", ctx->device);
    std::lock_guard<std::mutex> dev_lk(static_scratch_mu[ctx->device]);      // sibling contexts on this GPU share the scratch array
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane* L = nullptr;
    TRY(msm_submit<G1>(ctx, pts, off, n, d_scalars_mont, flags, &L, /*static_scratch=*/true));
    HG::X r;
    TRY(msm_collect<G1>(ctx, *L, &r));
    HG::write_jac(r, out_jac);
    return DVN_OK;
} DVN_CATCH

// B MultiExps over ONE point set in one pass of the pipeline: every (vector, window) pair is a virtual window of the same
// sorted entry list (msm_driver.h msm_windows_staged, batched mode), the window sums are combined per vector on the device.
// For many commitments over one basis (a PLONK prover's wire / quotient polynomials over one SRS, several BSB22 commitments
// over one key): B small MSMs cost one launch sequence instead of B.  scalars: batch x n images, vector b at [b*n, (b+1)*n).
extern "C" int32_t dvn_msm_g1_batched(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t batch,
                                      uint32_t flags, uint64_t* out_jac /* batch x 12 */) try {
    TRY(enter(ctx));
    if (!pts || !out_jac || (!scalars && n && batch)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->curve != 1 || pts->suite != 0) return fail(DVN_ERR_INVALID_ARG, "point set is not BN254 G1");
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "len(points) != len(scalars): range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    typedef dvn_host::HG1 HG;
    auto write_inf = [&](uint32_t b) { HG::write_jac(HG::infinity(), out_jac + (size_t)b * 12); };
    if (batch == 0) return DVN_OK;
    if (n == 0) { for (uint32_t b = 0; b < batch; ++b) write_inf(b); return DVN_OK; }
    u32 c, nwin;
    msm_plan(n, &c, &nwin);
    if ((size_t)batch * nwin * n >= ((size_t)1 << 31)) return fail(DVN_ERR_INVALID_ARG, "batched MSM too large: batch * windows * n must stay below 2^31");
    std::lock_guard<std::mutex> lk(ctx->mu);
    u32 *d_s = nullptr, *d_out = nullptr;
    const size_t sbytes = (size_t)batch * n * 32;
    auto cleanup = [&]() { if (d_s) (void)hipFree(d_s); if (d_out) (void)hipFree(d_out); };
    HIP_TRY(hipMalloc((void**)&d_out, (size_t)batch * G1::XW * 4));
    const u32* d_scalars = (const u32*)scalars;
    if (!(flags & DVN_MSM_SCALARS_ON_DEVICE)) {
        hipError_t e = hipMalloc((void**)&d_s, sbytes);
        if (e == hipSuccess) e = hipMemcpy(d_s, scalars, sbytes, hipMemcpyHostToDevice);        // caller's buffer is free on return
        if (e != hipSuccess) { cleanup(); (void)hipGetLastError(); return fail(DVN_ERR_OOM, "scalar upload failed: %s", hipGetErrorString(e)); }
        d_scalars = d_s;
    }
    MsmLane* L = nullptr;
    int32_t rc = msm_batched_submit<G1>(ctx, pts, off, n, d_scalars, batch, c, d_out, &L);
    std::vector<u32> h((size_t)batch * G1::XW);
    if (rc == DVN_OK) {
        hipError_t e = hipMemcpyAsync(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost, L->stream);
        rc = lane_wait(ctx, *L);
        if (rc == DVN_OK && e != hipSuccess) { (void)hipGetLastError(); rc = fail(DVN_ERR_DEVICE, "result copy failed: %s", hipGetErrorString(e)); }
    }
    cleanup();
    TRY(rc);
    for (uint32_t b = 0; b < batch; ++b) HG::write_jac(HG::load_device(h.data() + (size_t)b * G1::XW), out_jac + (size_t)b * 12);
    return DVN_OK;
} DVN_CATCH

// (window_bits, nwin, combine_bits) the library will use for an n-scalar MSM over this point set.  Expanded sets
// have pts-specific windows and combine_bits = 0: partial results of window ranges simply add up.
extern "C" int32_t dvn_msm_plan_points(const dvn_points* pts, size_t n, uint32_t* window_bits, uint32_t* nwin, uint32_t* combine_bits) try {
    if (!pts || !window_bits || !nwin || !combine_bits) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (use_expanded(pts, n)) { *window_bits = pts->expand_c; *nwin = pts->nexp; *combine_bits = 0; return DVN_OK; }
    msm_plan(n, window_bits, nwin);
    *combine_bits = *window_bits;
    return DVN_OK;
} DVN_CATCH
// window-sharded MSM, asynchronous halves: submit queues windows [win_lo, win_hi) on a free lane, collect exports the
// per-window sums (external-Montgomery X, Y, ZZ, ZZZ; infinity = all zero).  Expanded sets put the partial result in slot 0.
static int32_t windows_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags,
                              uint32_t win_lo, uint32_t win_hi, MsmLane** out) {
    if (!pts || (!scalars && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->curve != 1 || pts->suite != 0) return fail(DVN_ERR_INVALID_ARG, "point set is not BN254 G1");
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "len(points) != len(scalars)");
    const bool expanded = use_expanded(pts, n);
    u32 c, nwin;
    if (expanded) { c = pts->expand_c; nwin = pts->nexp; } else msm_plan(n, &c, &nwin);
    if (win_lo > win_hi || win_hi > nwin) return fail(DVN_ERR_INVALID_ARG, "window range [%u, %u) outside [0, %u)", win_lo, win_hi, nwin);
    MsmLane* L = nullptr;
    TRY(lane_acquire(ctx, &L));
    L->curve = 1; L->suite = 0; L->expanded = expanded; L->c = c; L->nwin = nwin; L->W = win_hi - win_lo;
    int32_t rc = DVN_OK;
    if (n && L->W) {
        rc = expanded ? msm_expanded<G1>(ctx, *L, pts, off, n, scalars, flags, win_lo, win_hi)
                      : msm_windows<G1>(ctx, *L, pts, off, n, scalars, flags, c, nwin, win_lo, win_hi);
    } else L->nwin = 0;                                     // nothing queued
    if (rc != DVN_OK) { (void)hipStreamSynchronize(L->stream); L->busy = false; return rc; }
    *out = L;
    return DVN_OK;
}
static int32_t windows_collect(dvn_ctx* ctx, MsmLane& L, uint64_t* window_sums) {
    const u32 W = L.W;
    std::vector<u32> sums((size_t)(W ? W : 1) * G1::XW, 0u);
    u32* o = (u32*)window_sums;
    u32 w0 = 0;
    if (L.nwin) {
        TRY(lane_wait(ctx, L));
        if (L.expanded) {                  // the pieces of the marginal-sum reduction -> ONE partial result in slot 0 (already R = 2^256 images)
            typedef dvn_host::HG1 HG;
            const HG::X r = bitsum_combine<HG>(L.h_sums, L.red_kh, L.red_kl, G1::XW);
            if (W) {
                if (HG::is_inf(r)) memset(o, 0, 128);
                else { memcpy(o, r.x.l, 32); memcpy(o + 8, r.y.l, 32); memcpy(o + 16, r.zz.l, 32); memcpy(o + 24, r.zzz.l, 32); }
                w0 = 1;
            }
        } else memcpy(sums.data(), L.h_sums, (size_t)W * G1::XW * 4);
    } else L.busy = false;
    for (u32 w = w0; w < W; ++w) {
        G1::X p = G1::load_xyzz_internal(sums.data() + (size_t)w * G1::XW);
        if (G1::is_inf(p)) { memset(o + (size_t)w * 32, 0, 128); continue; }
        Ops<Fp>::to_ext(p.x, o + w * 32); Ops<Fp>::to_ext(p.y, o + w * 32 + 8);
        Ops<Fp>::to_ext(p.zz, o + w * 32 + 16); Ops<Fp>::to_ext(p.zzz, o + w * 32 + 24);
    }
    return DVN_OK;
}
extern "C" int32_t dvn_msm_g1_windows_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags,
                                             uint32_t win_lo, uint32_t win_hi, int32_t* ticket) try {
    TRY(enter(ctx));
    if (!ticket) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane* L = nullptr;
    TRY(windows_submit(ctx, pts, off, n, scalars, flags, win_lo, win_hi, &L));
    *ticket = (int32_t)(L - ctx->lane) + 1;
    return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_msm_g1_windows_collect(dvn_ctx* ctx, int32_t ticket, uint64_t* window_sums) try {
    TRY(enter(ctx));
    if (!window_sums || ticket < 1 || ticket > DVN_LANES) return fail(DVN_ERR_INVALID_ARG, "bad ticket");
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane& L = ctx->lane[ticket - 1];
    if (!L.busy || L.curve != 1 || L.suite != 0) return fail(DVN_ERR_INVALID_ARG, "ticket %d is not an outstanding window-sharded BN254 G1 MSM", ticket);
    return windows_collect(ctx, L, window_sums);
} DVN_CATCH
extern "C" int32_t dvn_msm_g1_windows(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags,
                                      uint32_t win_lo, uint32_t win_hi, uint64_t* window_sums) try {
    TRY(enter(ctx));
    if (!window_sums) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane* L = nullptr;
    TRY(windows_submit(ctx, pts, off, n, scalars, flags, win_lo, win_hi, &L));
    return windows_collect(ctx, *L, window_sums);
} DVN_CATCH
extern "C" int32_t dvn_msm_combine_g1(dvn_ctx* ctx, const uint64_t* window_sums, uint32_t nwin, uint32_t window_bits, uint64_t out_jac[12]) try {
    (void)ctx;   /* pure host computation: ctx may be NULL (CPU-side tests of the sharded path) */
    if (!window_sums || !out_jac) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (window_bits > 24 || nwin == 0 || nwin > 64) return fail(DVN_ERR_INVALID_ARG, "bad window layout");   /* window_bits 0 = plain sum */
    const u32* in = (const u32*)window_sums;
    std::vector<u32> sums((size_t)nwin * G1::XW);
    for (u32 w = 0; w < nwin; ++w) {
        bool z = true;
        for (int i = 16; i < 24; ++i) z = z && in[w * 32 + i] == 0;
        G1::X p = z ? G1::infinity()
                    : G1::X{Ops<Fp>::from_ext(in + w * 32), Ops<Fp>::from_ext(in + w * 32 + 8), Ops<Fp>::from_ext(in + w * 32 + 16), Ops<Fp>::from_ext(in + w * 32 + 24)};
        G1::store_xyzz_internal(p, sums.data() + (size_t)w * G1::XW);
    }
    dvn_host::HG1::write_jac(dvn_host::HG1::horner(sums.data(), nwin, window_bits), out_jac);
    return DVN_OK;
} DVN_CATCH

// ---- one MSM shared between the GPUs of a node by entry count ---------------------------------------------------------------
// Rank `rank` of `world` computes ITS SHARE of sum s_i P_i and returns it as a G1Jac image; the shares of all ranks add up to
// the MSM (dvn_msm_g1_sum, host only).  For an expanded point set the share is an equal range of the nexp * n (row, point)
// entries: perfectly balanced for any world size, where whole rows are not (13 rows over 8 ranks).  The exchange between
// ranks is one all-gather of 96 bytes per rank (RCCL has no elliptic-curve reduction operator; the sum runs on the host).
// Scalars must already be in device memory.
extern "C" int32_t dvn_msm_g1_share_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* d_scalars_mont,
                                           uint32_t flags, uint32_t rank, uint32_t world, int32_t* ticket) try {
    TRY(enter(ctx));
    if (!pts || !ticket || (!d_scalars_mont && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->curve != 1 || pts->suite != 0) return fail(DVN_ERR_INVALID_ARG, "point set is not BN254 G1");
    if (!(flags & DVN_MSM_SCALARS_ON_DEVICE)) return fail(DVN_ERR_INVALID_ARG, "a shared MSM takes device scalars (DVN_MSM_SCALARS_ON_DEVICE)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane* L = nullptr;
    TRY(msm_share_submit<G1>(ctx, pts, off, n, (const u32*)d_scalars_mont, rank, world, &L));
    *ticket = (int32_t)(L - ctx->lane) + 1;
    return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_msm_g1_share_collect(dvn_ctx* ctx, int32_t ticket, uint64_t partial_jac[12]) try { return msm_collect_api<G1>(ctx, 1, ticket, partial_jac); } DVN_CATCH
extern "C" int32_t dvn_msm_g1_sum(const uint64_t* jacs, uint32_t count, uint64_t out_jac[12]) try {
    typedef dvn_host::HG1 HG;
    if ((!jacs && count) || !out_jac) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    HG::X acc = HG::infinity();
    for (uint32_t k = 0; k < count; ++k) {
        const u32* w = (const u32*)(jacs + (size_t)k * 12);
        bool z = true;
        for (int i = 0; i < 8; ++i) z = z && w[16 + i] == 0;                    // Z == 0: infinity
        if (z) continue;
        static const u64 ONE_MONT[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};   // 2^256 mod p
        if (memcmp(w + 16, ONE_MONT, 32) != 0) return fail(DVN_ERR_INVALID_ARG, "jacs[%u]: Z is neither 0 nor 1 -- this sum takes the normalised images the library emits (Z in {0, 1})", k);
        u32 img[G1::AW];
        G1::store_affine_internal(G1::affine_from_ext(w), img);                 // Z = 1 images (what the library emits)
        acc = HG::add(acc, HG::load_device_affine(img));
    }
    HG::write_jac(acc, out_jac);
    return DVN_OK;
} DVN_CATCH

// One MSM over ALL devices of a multi-device context: sets[i] is the same point set resident on device i.  A host thread per
// device runs that device's share (msm_share_submit with rank i of ndev: an equal range of the rows x n entries of the
// expanded set) on host scalars it uploads itself; the ndev partial sums are added on the host.  One process holds every
// partial result, so no collective runs (the one-rank-per-GPU form is dvn_msm_g1_share_submit + an all-gather of 96 bytes).
extern "C" int32_t dvn_msm_g1_multi(dvn_ctx* ctx, const dvn_points* const* sets, uint32_t nsets, size_t off, size_t n,
                                    const uint64_t* scalars_mont, uint64_t out_jac[12]) try {
    typedef dvn_host::HG1 HG;
    if (!ctx) return fail(DVN_ERR_INVALID_ARG, "ctx is NULL");
    if (!sets || !out_jac || (!scalars_mont && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const uint32_t ndev = 1 + (uint32_t)ctx->peers.size();
    if (nsets != ndev) return fail(DVN_ERR_INVALID_ARG, "%u point sets for a context of %u devices: one set per device", nsets, ndev);
    for (uint32_t i = 0; i < ndev; ++i) {
        const dvn_points* p = sets[i];
        dvn_ctx* ci = i == 0 ? ctx : ctx->peers[i - 1];
        if (!p) return fail(DVN_ERR_INVALID_ARG, "sets[%u] is NULL", i);
        if (p->curve != 1 || p->suite != 0) return fail(DVN_ERR_INVALID_ARG, "sets[%u] is not BN254 G1", i);
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p->d) != hipSuccess) { (void)hipGetLastError(); return fail(DVN_ERR_INVALID_ARG, "sets[%u] does not point at device memory", i); }
        if (at.device != ci->device) return fail(DVN_ERR_INVALID_ARG, "sets[%u] lives on device %d, the context's device %u is %d", i, at.device, i, ci->device);
        if (p->n != sets[0]->n || p->expand_c != sets[0]->expand_c || p->nexp != sets[0]->nexp)
            return fail(DVN_ERR_INVALID_ARG, "sets[%u] is not the same point set as sets[0] (length or table layout differs)", i);
        if (off > p->n || n > p->n - off) return fail(DVN_ERR_INVALID_ARG, "len(points) != len(scalars): range [%zu, %zu) outside point set of %zu", off, off + n, p->n);
    }
    std::vector<HG::X> parts(ndev, HG::infinity());
    std::vector<int32_t> rcs(ndev, DVN_OK);
    std::vector<std::string> errs(ndev);
    std::vector<std::thread> threads;
    threads.reserve(ndev);                                      // no reallocation (and no throw) once a thread runs
    bool spawn_failed = false;
    for (uint32_t i = 0; i < ndev && !spawn_failed; ++i) {
        try {
        threads.emplace_back([&, i]() {
            dvn_ctx* ci = i == 0 ? ctx : ctx->peers[i - 1];
            int32_t rc = enter(ci);                             // hipSetDevice on this thread
            if (rc == DVN_OK) {
                std::lock_guard<std::mutex> lk(ci->mu);
                MsmLane* L = nullptr;
                rc = msm_share_submit<G1>(ci, sets[i], off, n, (const u32*)scalars_mont, i, ndev, &L, DVN_MSM_DEFAULT);
                if (rc == DVN_OK) rc = msm_collect<G1>(ci, *L, &parts[i]);
            }
            rcs[i] = rc;
            if (rc != DVN_OK) errs[i] = tl_error;
        });
        } catch (...) { spawn_failed = true; }                  // no exception may cross the C boundary
    }
    for (auto& t : threads) t.join();
    if (spawn_failed) return fail(DVN_ERR_OOM, "could not start a host thread per device");
    for (uint32_t i = 0; i < ndev; ++i)
        if (rcs[i] != DVN_OK) return fail(rcs[i], "device share %u of %u failed: %s", i, ndev, errs[i].c_str());
    HG::X acc = HG::infinity();
    for (uint32_t i = 0; i < ndev; ++i) acc = HG::add(acc, parts[i]);
    HG::write_jac(acc, out_jac);
    return DVN_OK;
} DVN_CATCH

extern "C" size_t dvn_points_len(const dvn_points* pts) { return pts ? pts->n : 0; }

extern "C" int32_t dvn_points_free(dvn_ctx* ctx, dvn_points* pts) try {
    TRY(enter(ctx));
    if (!pts) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (pts->d) HIP_TRY(hipFree(pts->d));
    delete pts;
    return DVN_OK;
} DVN_CATCH

extern "C" int32_t dvn_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) try {
    TRY(enter(ctx));
    if (!pts || (!xy && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->suite != 0) return fail(DVN_ERR_INVALID_ARG, "point set is not a BN254 set (use the download call of its curve)");
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return pts->curve == 1 ? points_download<G1>(ctx, pts, off, n, xy) : dvn_internal_points_download_g2(ctx, pts, off, n, xy);
} DVN_CATCH

int32_t dvn_internal_points_set_expanded_g2(dvn_ctx* ctx, dvn_points* pts, u32 c);   // dvn_msm_g2.hip
extern "C" int32_t dvn_points_set_expanded(dvn_ctx* ctx, dvn_points* pts, uint32_t window_bits) try {
    TRY(enter(ctx));
    if (!pts) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return pts->curve == 1 ? points_set_expanded<G1>(ctx, pts, window_bits) : dvn_internal_points_set_expanded_g2(ctx, pts, window_bits);
} DVN_CATCH
extern "C" int32_t dvn_msm_plan(size_t n, uint32_t* window_bits, uint32_t* nwin) try {
    if (!window_bits || !nwin) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    msm_plan(n, window_bits, nwin);
    return DVN_OK;
} DVN_CATCH

extern "C" int32_t dvn_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out) try {
    TRY(enter(ctx));
    if (!d_out && n) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (kind != 0 && kind != 1) return fail(DVN_ERR_INVALID_ARG, "kind must be 0 (uniform) or 1 (witness-like)");
    if (n == 0) return DVN_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    k_scalars_synthetic<FrCfg><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((u32*)d_out, n, seed, kind);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DVN_OK;
} DVN_CATCH

int32_t dvn_internal_msm_g1_dev(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, dvn_host::HG1::X* out) {
    if (pts->curve != 1) return fail(DVN_ERR_INVALID_ARG, "point set is not G1");
    return msm_dev_xyzz<G1>(ctx, pts, off, n, d_scalars, out);
}

int32_t dvn_internal_msm_submit_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, MsmLane** lane) {
    if (pts->curve != 1) return fail(DVN_ERR_INVALID_ARG, "point set is not G1");
    return msm_submit<G1>(ctx, pts, off, n, (const uint64_t*)d_scalars, DVN_MSM_SCALARS_ON_DEVICE, lane);
}
int32_t dvn_internal_msm_collect_g1(dvn_ctx* ctx, MsmLane* lane, dvn_host::HG1::X* out) { return msm_collect<G1>(ctx, *lane, out); }
int32_t dvn_internal_msm_share_submit_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, u32 rank, u32 world, MsmLane** lane) {
    if (pts->curve != 1) return fail(DVN_ERR_INVALID_ARG, "point set is not G1");
    return msm_share_submit<G1>(ctx, pts, off, n, d_scalars, rank, world, lane);
}
