// Host drivers of the MSM pipeline, templated on the curve; included by dvn_msm_g1.hip / dvn_msm_g2.hip so the
// two instantiations compile in parallel.
#pragma once
#include "dvn_internal.h"
#include "msm_kernels.h"
#include "host_fast.h"

// scalar field of each group, and the finishing arithmetic used on the host
template <class Cv> struct ScalarOf;
template <> struct ScalarOf<G1> { typedef FrCfg S; };
template <> struct ScalarOf<G2> { typedef FrCfg S; };
template <> struct ScalarOf<G1_377> { typedef Fr377Cfg S; };
template <> struct ScalarOf<G2_377> { typedef Fr377Cfg S; };
template <class Cv> struct SuiteOf { static constexpr int value = 0; };          // 0 = BN254, 1 = BLS12-377
template <> struct SuiteOf<G1_377> { static constexpr int value = 1; };
template <> struct SuiteOf<G2_377> { static constexpr int value = 1; };
template <> struct ScalarOf<G1_381> { typedef Fr381Cfg S; };
template <> struct ScalarOf<G2_381> { typedef Fr381Cfg S; };
template <> struct SuiteOf<G1_381> { static constexpr int value = 3; };               // 3 = BLS12-381
template <> struct SuiteOf<G2_381> { static constexpr int value = 3; };
template <> struct ScalarOf<G_761> { typedef Fp377Cfg S; };                          // BW6-761's scalar field is BLS12-377's base field
template <> struct SuiteOf<G_761> { static constexpr int value = 2; };               // 2 = BW6-761
template <class F> struct IsQuadratic { static constexpr bool value = false; };
template <class C> struct IsQuadratic<Fe2<C>> { static constexpr bool value = true; };

// ------------------------------------------------------------------------------------------------------
// points
// ------------------------------------------------------------------------------------------------------
template <class Cv> static int32_t points_upload(dvn_ctx* ctx, const uint64_t* xy, size_t n, int curve, dvn_points** out) {
    TRY(enter(ctx));
    if (!out || (!xy && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (n >= (1ull << 31)) return fail(DVN_ERR_INVALID_ARG, "point set too large (n must be < 2^31)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = curve; p->n = n;
    p->suite = SuiteOf<Cv>::value;
    expand_plan(n, ScalarOf<Cv>::S::BITS, (size_t)Cv::AW * 4, &p->expand_c, &p->nexp);
    size_t bytes = n * Cv::AW * 4;
    hipError_t e = hipMalloc((void**)&p->d, bytes ? bytes * p->nexp : 16);
    if (e != hipSuccess) { delete p; (void)hipGetLastError(); return fail(DVN_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    if (n) {
        e = hipMemcpyAsync(p->d, xy, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            k_points_to_internal<Cv><<<(unsigned)((n + 127) / 128), 128, 0, ctx->stream>>>(p->d, n);
            if (p->expand_c) k_points_expand<Cv><<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(p->d, n, p->expand_c, p->nexp);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // caller's buffer is free after return
        if (e != hipSuccess) { (void)hipFree(p->d); delete p; (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "point upload failed: %s", hipGetErrorString(e)); }
    }
    *out = p;
    return DVN_OK;
}

// P_i = base + i*step on the device (bench/test input generator; caller holds ctx->mu)
template <class Cv> static int32_t points_progression(dvn_ctx* ctx, size_t n, int curve, const typename Cv::A& base, const typename Cv::A& step, dvn_points** out) {
    u32 hb[2 * Cv::AW];
    Cv::store_affine_internal(base, hb); Cv::store_affine_internal(step, hb + Cv::AW);
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = curve; p->n = n;
    p->suite = SuiteOf<Cv>::value;
    expand_plan(n, ScalarOf<Cv>::S::BITS, (size_t)Cv::AW * 4, &p->expand_c, &p->nexp);
    u32* dbs = nullptr;
    hipError_t e = hipMalloc((void**)&p->d, n ? n * Cv::AW * 4 * p->nexp : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&dbs, sizeof(hb));
    if (e == hipSuccess) e = hipMemcpyAsync(dbs, hb, sizeof(hb), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) {
        const u32 chunk = 16;
        size_t threads = (n + chunk - 1) / chunk;
        k_points_progression<Cv><<<(unsigned)((threads + 63) / 64), 64, 0, ctx->stream>>>(p->d, n, dbs, chunk);
        if (p->expand_c) k_points_expand<Cv><<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(p->d, n, p->expand_c, p->nexp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (dbs) (void)hipFree(dbs);
    if (e != hipSuccess) { if (p->d) (void)hipFree(p->d); delete p; (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "synthetic points failed: %s", hipGetErrorString(e)); }
    *out = p;
    return DVN_OK;
}

template <class Cv> static int32_t points_synthetic(dvn_ctx* ctx, uint64_t seed, size_t n, int curve, const typename Cv::A& gen, dvn_points** out) {
    TRY(enter(ctx));
    if (!out) return fail(DVN_ERR_INVALID_ARG, "out is NULL");
    if (n >= (1ull << 31)) return fail(DVN_ERR_INVALID_ARG, "point set too large");
    std::lock_guard<std::mutex> lk(ctx->mu);
    u32 s0[8], s1[8]; fr_at_words(seed, 0, s0); fr_at_words(seed, 1, s1);
    typename Cv::A base = Cv::to_affine(host_scalar_mul<Cv>(gen, s0)), step = Cv::to_affine(host_scalar_mul<Cv>(gen, s1));
    u32 hb[2 * Cv::AW];
    Cv::store_affine_internal(base, hb); Cv::store_affine_internal(step, hb + Cv::AW);
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = curve; p->n = n;
    p->suite = SuiteOf<Cv>::value;
    expand_plan(n, ScalarOf<Cv>::S::BITS, (size_t)Cv::AW * 4, &p->expand_c, &p->nexp);
    u32* dbs = nullptr;
    hipError_t e = hipMalloc((void**)&p->d, n ? n * Cv::AW * 4 * p->nexp : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&dbs, sizeof(hb));
    if (e == hipSuccess) e = hipMemcpyAsync(dbs, hb, sizeof(hb), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n) {
        const u32 chunk = 16;
        size_t threads = (n + chunk - 1) / chunk;
        k_points_progression<Cv><<<(unsigned)((threads + 63) / 64), 64, 0, ctx->stream>>>(p->d, n, dbs, chunk);
        if (p->expand_c) k_points_expand<Cv><<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(p->d, n, p->expand_c, p->nexp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (dbs) (void)hipFree(dbs);
    if (e != hipSuccess) { if (p->d) (void)hipFree(p->d); delete p; (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "synthetic points failed: %s", hipGetErrorString(e)); }
    *out = p;
    return DVN_OK;
}

template <class Cv> static int32_t points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy) {
    if (n == 0) return DVN_OK;
    u32* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, n * Cv::AW * 4));
    k_points_to_external<Cv><<<(unsigned)((n + 127) / 128), 128, 0, ctx->stream>>>(pts->d + off * Cv::AW, tmp, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(xy, tmp, n * Cv::AW * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "point download failed: %s", hipGetErrorString(e)); }
    return DVN_OK;
}

// ------------------------------------------------------------------------------------------------------
// MSM
// ------------------------------------------------------------------------------------------------------
static void msm_plan(size_t n, u32* c_out, u32* nwin_out, u32 bits = 254) {
    // minimise nwin * (n + 3 * 2^(c-1)): n mixed additions per window plus ~3 additions per bucket to reduce
    u32 best = 8; double bc = 1e300;
    for (u32 c = 6; c <= 16; ++c) {
        double cost = (double)(bits / c + 1) * ((double)n + 3.0 * (double)(1u << (c - 1)));
        if (cost < bc) { bc = cost; best = c; }
    }
    *c_out = best; *nwin_out = bits / best + 1;
}

// Segment length for k_accumulate: every lane executes exactly `seg` additions.  Measured (tools/gpu_seg.py, 2^22):
// anything in 32..128 is within 3 %, longer segments lose to the tail of the last wave round, shorter ones to the
// edge partials -- aim at >= 4 rounds of the resident lanes, clamp to [16, 128].
// Kernel curve type: Fp2 curves run every group-arithmetic kernel of the MSM with one Fp2 component per lane (fp2_paired.h).
// The memory images are identical, so host code and the point-set kernels keep the ordinary Curve<Fe2<C>>.
template <class Cv> struct PairedOf { typedef Cv type; };
template <class C> struct PairedOf<Curve<Fe2<C>>> { typedef Curve<Fe2P<C>> type; };
template <class Cv> static u32 pick_segment(size_t total) {
    static size_t slots = 0;                         // resident segments of the accumulate kernel on this device
    if (!slots) {
        int blocks = 0, dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void*)k_accumulate<typename PairedOf<Cv>::type>, 128, 0) == hipSuccess && blocks > 0)
            slots = (size_t)prop.multiProcessorCount * blocks * 128 / Lanes<typename PairedOf<Cv>::type>::LPP;
        else { (void)hipGetLastError(); slots = 196608; }
    }
    static const int forced = []() { const char* v = getenv("DVN_MSM_SEG"); return v ? atoi(v) : 0; }();
    if (forced >= 2) return (u32)forced;
    size_t seg = total / (slots * 4);
    if (seg < 16) seg = 16;
    if (seg > 128) seg = 128;
    if (seg > 48 && seg < 96) seg = 64;              // sweep at 2^22 x 13 rows: 64 is 3-7 % faster than its neighbours 56..80
    return (u32)seg;
}

// window sums for windows [win_lo, win_hi) -> host, internal packed xyzz images (XW words each)
// ENQUEUE only: the kernels and the D2H copy of the window sums (into L.h_sums) are queued on the lane's stream;
// lane_wait() completes them.
// sliced reduction of giant buckets (k_giant_*): on for every curve
template <class Cv> struct GiantOk { static constexpr bool value = true; };
static inline size_t acc_lds_probe() { static const size_t v = []() { const char* e = getenv("DVN_ACC_LDS"); return e ? (size_t)atol(e) : 0; }(); return v; }   // occupancy probe
static inline u32 giant_threshold() {       // DVN_MSM_GIANT=<segments> overrides (debug / A-B); never below FIX_GIANT: giant_cap assumes it
    static const u32 v = []() { const char* e = getenv("DVN_MSM_GIANT"); const long x = e ? atol(e) : 0; return x > (long)FIX_GIANT ? (u32)x : FIX_GIANT; }();
    return v;
}
// LDS limits of the staged sort kernels (dynamic LDS above the 64 KiB default needs the attribute once per kernel)
// (a function attribute is a property of the function ON ONE DEVICE: once per device, with that device current)
template <class D> static int32_t sort_attrs(int device) {
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64) return fail(DVN_ERR_INVALID_ARG, "device ordinal %d out of range", device);
    if (!done[device]) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_s1_scatter<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)k_s2_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        done[device] = true;
    }
    return DVN_OK;
}
constexpr u32 S1_TPB = 1;                     // tiles per pass-1 block of an expanded set's sort
template <class S> static int32_t digits_hist_attr(int device) {
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64) return fail(DVN_ERR_INVALID_ARG, "device ordinal %d out of range", device);
    if (!done[device]) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_digits32_hist<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        done[device] = true;
    }
    return DVN_OK;
}
static inline size_t s1_lds(u32 lb) { return ((size_t)3 * lb + 80 + 2 * ST) * 4; }
static inline size_t s2_lds() { return ((size_t)4 * 1024 + 80 + ST) * 4 + (size_t)ST * 2; }

// zero `bytes` (a multiple of 4) on the lane's stream: always a fill kernel of small workgroups (dvn_zero_async), never the
// runtime's memset -- that one cannot start beside another lane's accumulation, and it is not to be recorded under graph capture
static inline int32_t lane_zero(MsmLane& L, void* p, size_t bytes, hipStream_t st) { (void)L; return dvn_zero_async(p, bytes, st); }

// Plain point sets, two-pass staged sort: one globally sorted entry list over the virtual buckets (window, bucket);
// the accumulate / fix kernels see ONE window of W * nb buckets, the reduction runs per real window afterwards.
template <class Cv>
static int32_t msm_windows_staged(dvn_ctx* ctx, MsmLane& L, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags,
                                  u32 c, u32 win_lo, u32 win_hi, u32 batch = 1, u32* d_batch_out = nullptr, const u32* row_table = nullptr) {
    // d_batch_out != NULL: `scalars` holds batch x n device scalars (vector b at [b * n, (b + 1) * n)) over the SAME n points; every
    // (vector, window) pair is a virtual window of the one sorted entry list, the window sums are combined per vector on
    // the device (k_batch_horner) and the batch internal xyzz results stay in d_batch_out (device).
    typedef typename PairedOf<Cv>::type KC;
    constexpr u32 LPP = Lanes<KC>::LPP;
    constexpr size_t AW_ = AccImg<KC>::W;                 // words per raw accumulator image (bucket sums, edge partials)
    static_assert(KC::XW == Cv::XW && KC::AW == Cv::AW, "paired and ordinary images must agree");
    // row_table != NULL (batched only): the points are an expanded table, row w = 2^(c w) * P_i (n points per row, win_hi
    // rows).  The digit array has the same memory layout, but is read as ONE virtual window per vector with win_hi * n
    // entries whose point reference is the table position w * n + i: all windows of a vector share one bucket set, there
    // is one bucket reduction per vector and no window combine (carried over to the batched path).
    const bool batched = d_batch_out != nullptr;
    const bool rowed = row_table != nullptr;
    if ((batch != 1 && !batched) || (rowed && !batched) || (batched && (win_lo != 0 || batch == 0 || !(flags & DVN_MSM_SCALARS_ON_DEVICE)))) return fail(DVN_ERR_INVALID_ARG, "batched MSM: all windows, device scalars, device output");
    const size_t n_src = n;                                            // scalars per vector
    if (rowed) n = n_src * win_hi;                                     // entries per virtual window
    const u32 W = rowed ? batch : (win_hi - win_lo) * batch;
    const u32 nb = 1u << (c - 1);
    const u32 lob = c - 1 < 7 ? c - 1 : 7, lb = nb >> lob;             // <= 256 bins per window, 2^lob sub-buckets per bin
    const size_t total = (size_t)W * n;
    if (total >= ((size_t)1 << 31)) return fail(DVN_ERR_INVALID_ARG, "MSM too large for 31-bit entry positions");
    const u32 vb = W * nb;                                             // virtual buckets
    const u32 seg = pick_segment<Cv>(total);
    const u32 nseg = (u32)((total + seg - 1) / seg);
    const u32 LB8 = 8;
    const u32 bpw = (nb + 256 * LB8 - 1) / (256 * LB8);
    const u32 heavy_cap = (u32)(nseg / (FIX_LIGHT - 1) + 1);
    const u32 giant_cap = heavy_cap / (FIX_GIANT / FIX_LIGHT) + 1;
    const u32 tpw = (u32)((n + ST - 1) / ST);
    const u32 hbins = W * lb;
    const bool on_dev = flags & DVN_MSM_SCALARS_ON_DEVICE;

    CarverSize cs;
    constexpr size_t SB = ScalarOf<Cv>::S::EW * 4;       // bytes per scalar
    if (!on_dev) cs.take(n * SB);
    cs.take(total * 2); cs.take(total * 4); cs.take(total * 2); cs.take(total * 4);      // digits, p1_ref, p1_lo, sorted
    const u32 nblk = (tpw + S1_RPB - 1) / S1_RPB;                      // row blocks of the prefix pass over hist1, per window
    cs.take((size_t)W * tpw * lb * 4); cs.take((size_t)(3 * hbins + 4) * 4);             // hist1; bin_total, bin_base, piece_base
    cs.take((size_t)W * nblk * lb * 4);                                                  // part
    cs.take((size_t)vb * 4 * 3); cs.take(16);                                            // counts, cursor, offsets
    cs.take((size_t)heavy_cap * 4);
    cs.take((size_t)giant_cap * 4); cs.take((size_t)giant_cap * GIANT_PIECES * Cv::XW * 4);
    cs.take((size_t)vb * AW_ * 4);
    cs.take((size_t)nseg * AW_ * 4); cs.take((size_t)nseg * AW_ * 4);
    cs.take((size_t)W * bpw * Cv::XW * 4); cs.take((size_t)W * Cv::XW * 4);
    TRY(lane_ensure_ws(L, cs.off + 8192));
    Carver cv(lane_ws_base(L));
    const u32* d_scalars;
    if (!on_dev) {
        u32* tmp = cv.take<u32>(n * (SB / 4));
        TRY(dvn_h2d(ctx, tmp, scalars, n * SB, L.stream));                 // caller's buffer is free on return (cgo rule)
        d_scalars = tmp;
    } else d_scalars = (const u32*)scalars;
    int16_t* digits = cv.take<int16_t>(total);
    u32* p1_ref = cv.take<u32>(total);
    unsigned short* p1_lo = cv.take<unsigned short>(total);
    u32* sorted = cv.take<u32>(total);
    u32* hist1 = cv.take<u32>((size_t)W * tpw * lb);
    u32* bin_total = cv.take<u32>(3 * hbins + 4);
    u32* part = cv.take<u32>((size_t)W * nblk * lb);
    u32* bin_base = bin_total + hbins;                                 // hbins + 1 entries
    u32* piece_base = bin_base + hbins + 1;                            // hbins + 1 entries
    u32* counts = cv.take<u32>((size_t)vb * 3);
    u32* cursor = counts + vb;
    u32* offsets = cursor + vb;
    u32* totals = cv.take<u32>(4);
    u32* heavy_count = totals + 1;
    u32* giant_count = totals + 2;
    u32* heavy_list = cv.take<u32>(heavy_cap);
    u32* giant_list = cv.take<u32>(giant_cap);
    u32* giant_part = cv.take<u32>((size_t)giant_cap * GIANT_PIECES * Cv::XW);
    u32* buckets = cv.take<u32>((size_t)vb * AW_);
    const u32 max_pieces = (u32)(total / ST) + hbins;
    u32* edge_first = cv.take<u32>((size_t)nseg * AW_);
    u32* edge_last = cv.take<u32>((size_t)nseg * AW_);
    u32* partials = cv.take<u32>((size_t)W * bpw * Cv::XW);
    u32* wsums = cv.take<u32>((size_t)W * Cv::XW);

    hipStream_t st = L.stream;
    static const bool dbg = getenv("DVN_DEBUG_SYNC") != nullptr;
#define DBG_SYNC(name) do { if (dbg) { hipError_t _e = hipStreamSynchronize(st); fprintf(stderr, "[dvn] %s: %s\n", name, hipGetErrorString(_e)); fflush(stderr); if (_e != hipSuccess) return fail(DVN_ERR_DEVICE, "%s: %s", name, hipGetErrorString(_e)); } } while (0)
    if (dbg) { fprintf(stderr, "[dvn] msm(staged) n=%zu c=%u W=%u nb=%u lob=%u lb=%u seg=%u nseg=%u tpw=%u ws=%zu\n", n, c, W, nb, lob, lb, seg, nseg, tpw, cs.off); fflush(stderr); }
    TRY(sort_attrs<int16_t>(ctx->device));
    HIP_TRY(hipEventRecord(L.ev[0], st));
    TRY(lane_zero(L, totals, 16, st));
    if (batched) k_digits16_batched<typename ScalarOf<Cv>::S><<<(unsigned)((n_src * batch + 255) / 256), 256, 0, st>>>(d_scalars, n_src, batch, c, win_hi, digits);
    else k_digits16<typename ScalarOf<Cv>::S><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_scalars, n, c, win_lo, win_hi, digits);
    DBG_SYNC("k_digits16");
    k_s1_hist<int16_t><<<W * tpw, 1024, (size_t)lb * 4, st>>>(digits, n, tpw, 1u, lob, lb, hist1);
    DBG_SYNC("k_s1_hist");
    k_s1_offsets_a<<<W * nblk, lb < 1024 ? (lb < 64 ? 64 : lb) : 1024, 0, st>>>(hist1, part, lb, tpw, nblk);
    k_s1_offsets_b<<<W, lb < 1024 ? (lb < 64 ? 64 : lb) : 1024, 0, st>>>(part, bin_total, lb, nblk);
    DBG_SYNC("k_s1_offsets");
    k_p1_scan<<<1, 1024, 0, st>>>(bin_total, bin_base, totals, hbins);
    DBG_SYNC("k_p1_scan");
    k_s1_scatter<int16_t><<<W * tpw, 1024, s1_lds(lb), st>>>(digits, n, tpw, 1u, 1u, lob, lb, 0u, 0u, win_lo, hist1, part, nblk, bin_base, p1_ref, p1_lo);
    DBG_SYNC("k_s1_scatter");
    TRY(lane_zero(L, counts, (size_t)vb * 8, st));             // counts + cursor
    k_s2_pieces<<<1, 1024, 0, st>>>(bin_base, hbins, piece_base);
    DBG_SYNC("k_s2_pieces");
    k_s2_count<<<max_pieces, 1024, 0, st>>>(bin_base, piece_base, hbins, p1_lo, lob, counts);
    DBG_SYNC("k_s2_count");
    k_s2_scatter<<<max_pieces, 1024, s2_lds(), st>>>(bin_base, piece_base, hbins, p1_ref, p1_lo, lob, counts, cursor, sorted, offsets);
    DBG_SYNC("k_s2_scatter");
    HIP_TRY(hipEventRecord(L.ev[2], st));
    k_accumulate<KC><<<(unsigned)(((size_t)nseg * LPP + 127) / 128), 128, acc_lds_probe(), st>>>(rowed ? row_table : pts->d, rowed ? 0 : off, sorted, offsets, counts, totals, buckets,
                                                                          edge_first, edge_last, total, vb, seg, nseg, 1);
    DBG_SYNC("k_accumulate");
    HIP_TRY(hipEventRecord(L.ev[3], st));
    k_fix<KC><<<(unsigned)(((size_t)vb * LPP + 127) / 128), 128, 0, st>>>(offsets, counts, buckets, edge_first, edge_last, heavy_list,
                                                                 heavy_count, 1, vb, seg, nseg, heavy_cap);
    DBG_SYNC("k_fix");
    const size_t sh_bytes = (size_t)256 * Cv::XW * 4;
    k_fix_heavy<KC><<<512, 256 * LPP, sh_bytes, st>>>(offsets, counts, buckets, edge_first, edge_last, heavy_list, heavy_count, vb, seg, nseg, heavy_cap,
                                                giant_list, giant_count, giant_cap, GiantOk<Cv>::value ? giant_threshold() : 0xffffffffu);
    DBG_SYNC("k_fix_heavy");
    if constexpr (GiantOk<Cv>::value) {
        k_giant_partial<KC><<<1024, 256 * LPP, sh_bytes, st>>>(offsets, counts, edge_first, edge_last, giant_list, giant_count, giant_cap, vb, seg, nseg, giant_part);
        k_giant_final<KC><<<64, GIANT_PIECES * LPP, (size_t)GIANT_PIECES * Cv::XW * 4, st>>>(giant_list, giant_count, giant_cap, giant_part, buckets);
        DBG_SYNC("k_giant");
    }
    k_bucket_reduce<KC><<<W * bpw, 256 * LPP, sh_bytes, st>>>(buckets, partials, nb, LB8, bpw);
    DBG_SYNC("k_bucket_reduce");
    k_window_finalize<KC><<<W, 64 * LPP, (size_t)64 * Cv::XW * 4, st>>>(partials, wsums, W, bpw);
    DBG_SYNC("k_window_finalize");
    if (rowed) {
        HIP_TRY(hipMemcpyAsync(d_batch_out, wsums, (size_t)W * Cv::XW * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipEventRecord(L.ev[1], st));
        return DVN_OK;
    }
    if (batched) {
        k_batch_horner<Cv><<<(batch + 63) / 64, 64, 0, st>>>(wsums, win_hi, c, batch, d_batch_out);
        DBG_SYNC("k_batch_horner");
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(L.ev[1], st));
        return DVN_OK;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(L.ev[1], st));
    if ((size_t)W * Cv::XW > DVN_HSUMS_WORDS) return fail(DVN_ERR_INVALID_ARG, "too many windows for the result buffer");
    HIP_TRY(hipMemcpyAsync(L.h_sums, wsums, (size_t)W * Cv::XW * 4, hipMemcpyDeviceToHost, st));
    return DVN_OK;
}

template <class Cv>
static int32_t msm_windows(dvn_ctx* ctx, MsmLane& L, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags,
                           u32 c, u32 nwin, u32 win_lo, u32 win_hi) {
    (void)nwin;
    return msm_windows_staged<Cv>(ctx, L, pts, off, n, scalars, flags, c, win_lo, win_hi);
}

// --------------------------------------------------------------------------------------------------------------------
// Expanded-table MSM (msm_kernels.h "Expanded point sets"): virtual windows [win_lo, win_hi) of the pts->nexp windows of
// width pts->expand_c, all accumulated into ONE bucket set.  host_sum receives ONE internal xyzz image: the partial
// result of these windows, window weights included (partial results of different window ranges simply add up).
// --------------------------------------------------------------------------------------------------------------------
// Pieces the reduction of an expanded bucket set leaves for the host (k_marginal_bitsums): layout and combine in host_fast.h
using dvn_host::bitsum_pieces;
template <class HG> static typename HG::X bitsum_combine(const u32* h, u32 kh, u32 kl, u32 xw) {
    return dvn_host::bitsum_combine_pieces<HG>([&](size_t i) { return HG::load_device(h + i * xw); }, kh, kl);
}
static inline bool use_expanded(const dvn_points* pts, size_t n) {
    static const u32 min_log = [] { const char* e = getenv("DVN_MSM_EXPANDED_MIN_LOG"); const int v = e ? atoi(e) : 0; return (u32)(v >= 8 && v <= 20 ? v : DVN_EXPAND_MIN_LOG); }();
    return pts->expand_c != 0 && n >= ((size_t)1 << min_log) && n * 4 >= pts->n;
}
template <class Cv>
static int32_t msm_expanded(dvn_ctx* ctx, MsmLane& L, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags,
                            u32 win_lo, u32 win_hi, u64 flat_lo = 0, u64 flat_hi = ~(u64)0) {
    typedef typename PairedOf<Cv>::type KC;
    constexpr u32 LPP = Lanes<KC>::LPP;
    constexpr size_t AW_ = AccImg<KC>::W;                 // words per raw accumulator image (bucket sums, edge partials)
    const u32 c = pts->expand_c, W = win_hi - win_lo;
    const u32 nb = 1u << (c - 1);
    const u32 lob = (c - 1 + 1) / 2 < 10 ? (c - 1 + 1) / 2 : 10, lb = nb >> lob;        // balanced split of the bucket bits
    if (lb > 1024) return fail(DVN_ERR_INVALID_ARG, "expanded window size %u too large for the two-pass sort", c);
    const size_t total = (size_t)W * n;                              // upper bound on the number of entries
    if (total >= ((size_t)1 << 31) || (size_t)pts->nexp * pts->n >= ((size_t)1 << 31)) return fail(DVN_ERR_INVALID_ARG, "MSM too large for 31-bit entry references");
    const u32 seg = pick_segment<Cv>(total);
    const u32 nseg = (u32)((total + seg - 1) / seg);
    // bucket reduction by marginal sums (msm_kernels.h k_marginal_level): nb = RH rows x RL columns
    const u32 kl = c / 2, kh = c - 1 - kl, RL = 1u << kl, RH = 1u << kh;
    const u32 npieces = bitsum_pieces(kh, kl);
    if ((size_t)npieces * Cv::XW > DVN_HSUMS_WORDS) return fail(DVN_ERR_INVALID_ARG, "too many reduction pieces for the result buffer");
    const u32 heavy_cap = (u32)(nseg / (FIX_LIGHT - 1) + 1);
    const u32 giant_cap = heavy_cap / (FIX_GIANT / FIX_LIGHT) + 1;
    const u32 tpw = (u32)((n + ST - 1) / ST);
    const u32 gpw = (tpw + S1_TPB - 1) / S1_TPB;                     // pass 1 of the sort works on groups of S1_TPB tiles
    if ((size_t)W * lb * 4 > 96 * 1024) return fail(DVN_ERR_INVALID_ARG, "expanded window layout needs %zu bytes of LDS histograms", (size_t)W * lb * 4);
    const u32 max_pieces = (u32)(total / ST) + lb;
    const bool on_dev = flags & DVN_MSM_SCALARS_ON_DEVICE;

    CarverSize cs;
    constexpr size_t SB = ScalarOf<Cv>::S::EW * 4;       // bytes per scalar
    if (!on_dev) cs.take(n * SB);
    cs.take(total * 4); cs.take(total * 4); cs.take(total * 2); cs.take(total * 4);    // digits, p1_ref, p1_lo, sorted
    const u32 nblk = (W * gpw + S1_RPB - 1) / S1_RPB;                // row blocks of the prefix pass over hist1
    cs.take((size_t)W * gpw * lb * 4); cs.take((size_t)nblk * lb * 4); cs.take((size_t)(3 * lb + 4) * 4);                // hist1; bin_total, bin_base, piece_base
    cs.take((size_t)nb * 4 * 3); cs.take(16);                                            // counts, cursor, offsets; totals + heavy_count
    cs.take((size_t)heavy_cap * 4);
    cs.take((size_t)giant_cap * 4); cs.take((size_t)giant_cap * GIANT_PIECES * Cv::XW * 4);
    cs.take((size_t)nb * AW_ * 4);
    cs.take((size_t)nseg * AW_ * 4); cs.take((size_t)nseg * AW_ * 4);
    cs.take((size_t)(RH + RL) * Cv::XW * 4); cs.take((size_t)npieces * Cv::XW * 4);
    cs.take((size_t)(nb / 2) * Cv::XW * 4); cs.take((size_t)(nb / 4 + 2) * Cv::XW * 4);    // marginal levels, ping and pong
    TRY(lane_ensure_ws(L, cs.off + 8192));
    Carver cv(lane_ws_base(L));
    const u32* d_scalars;
    if (!on_dev) {
        u32* tmp = cv.take<u32>(n * (SB / 4));
        TRY(dvn_h2d(ctx, tmp, scalars, n * SB, L.stream));                 // caller's buffer is free on return (cgo rule)
        d_scalars = tmp;
    } else d_scalars = (const u32*)scalars;
    int* digits = cv.take<int>(total);
    u32* p1_ref = cv.take<u32>(total);
    unsigned short* p1_lo = cv.take<unsigned short>(total);
    u32* sorted = cv.take<u32>(total);
    u32* hist1 = cv.take<u32>((size_t)W * gpw * lb);
    u32* part = cv.take<u32>((size_t)nblk * lb);
    u32* bin_total = cv.take<u32>(3 * lb + 4);
    u32* bin_base = bin_total + lb;
    u32* piece_base = bin_base + lb + 1;
    u32* counts = cv.take<u32>((size_t)nb * 3);
    u32* cursor = counts + nb;
    u32* offsets = cursor + nb;
    u32* totals = cv.take<u32>(4);
    u32* heavy_count = totals + 1;
    u32* giant_count = totals + 2;
    u32* heavy_list = cv.take<u32>(heavy_cap);
    u32* giant_list = cv.take<u32>(giant_cap);
    u32* giant_part = cv.take<u32>((size_t)giant_cap * GIANT_PIECES * Cv::XW);
    u32* buckets = cv.take<u32>((size_t)nb * AW_);
    u32* edge_first = cv.take<u32>((size_t)nseg * AW_);
    u32* edge_last = cv.take<u32>((size_t)nseg * AW_);
    u32* marg = cv.take<u32>((size_t)(RH + RL) * Cv::XW);
    u32* pieces = cv.take<u32>((size_t)npieces * Cv::XW);
    u32* level_buf[2] = {cv.take<u32>((size_t)(nb / 2) * Cv::XW), cv.take<u32>((size_t)(nb / 4 + 2) * Cv::XW)};

    hipStream_t st = L.stream;
    static const bool dbg = getenv("DVN_DEBUG_SYNC") != nullptr;
    if (dbg) { fprintf(stderr, "[dvn] msm(expanded) n=%zu c=%u W=%u nb=%u lob=%u lb=%u seg=%u nseg=%u tpw=%u ws=%zu\n", n, c, W, nb, lob, lb, seg, nseg, tpw, cs.off); fflush(stderr); }
    TRY(sort_attrs<int>(ctx->device));
    TRY(digits_hist_attr<typename ScalarOf<Cv>::S>(ctx->device));
    HIP_TRY(hipEventRecord(L.ev[0], st));
    TRY(lane_zero(L, totals, 16, st));
    k_digits32_hist<typename ScalarOf<Cv>::S><<<gpw, 1024, (size_t)W * lb * 4, st>>>(d_scalars, n, c, win_lo, win_hi, digits, flat_lo, flat_hi, gpw, S1_TPB, lob, lb, hist1);
    DBG_SYNC("k_digits32_hist");
    k_s1_offsets_a<<<nblk, lb < 1024 ? (lb < 64 ? 64 : lb) : 1024, 0, st>>>(hist1, part, lb, W * gpw, nblk);
    k_s1_offsets_b<<<1, lb < 1024 ? (lb < 64 ? 64 : lb) : 1024, 0, st>>>(part, bin_total, lb, nblk);
    DBG_SYNC("k_s1_offsets");
    k_p1_scan<<<1, 1024, 0, st>>>(bin_total, bin_base, totals, lb);
    DBG_SYNC("k_p1_scan");
    k_s1_scatter<int><<<W * gpw, 1024, s1_lds(lb), st>>>(digits, n, gpw, S1_TPB, 0u, lob, lb, (u32)off, (u32)pts->n, win_lo, hist1, part, nblk, bin_base, p1_ref, p1_lo);
    DBG_SYNC("k_s1_scatter");
    TRY(lane_zero(L, counts, (size_t)nb * 8, st));             // counts + cursor
    k_s2_pieces<<<1, 1024, 0, st>>>(bin_base, lb, piece_base);
    DBG_SYNC("k_s2_pieces");
    k_s2_count<<<max_pieces, 1024, 0, st>>>(bin_base, piece_base, lb, p1_lo, lob, counts);
    DBG_SYNC("k_s2_count");
    k_s2_scatter<<<max_pieces, 1024, s2_lds(), st>>>(bin_base, piece_base, lb, p1_ref, p1_lo, lob, counts, cursor, sorted, offsets);
    DBG_SYNC("k_s2_scatter");
    HIP_TRY(hipEventRecord(L.ev[2], st));
    k_accumulate<KC><<<(unsigned)(((size_t)nseg * LPP + 127) / 128), 128, acc_lds_probe(), st>>>(pts->d, 0, sorted, offsets, counts, totals, buckets,
                                                                          edge_first, edge_last, total, nb, seg, nseg, 1);
    DBG_SYNC("k_accumulate");
    HIP_TRY(hipEventRecord(L.ev[3], st));
    k_fix<KC><<<(unsigned)(((size_t)nb * LPP + 127) / 128), 128, 0, st>>>(offsets, counts, buckets, edge_first, edge_last, heavy_list,
                                                                 heavy_count, 1, nb, seg, nseg, heavy_cap);
    DBG_SYNC("k_fix");
    const size_t sh_bytes = (size_t)256 * Cv::XW * 4;
    k_fix_heavy<KC><<<512, 256 * LPP, sh_bytes, st>>>(offsets, counts, buckets, edge_first, edge_last, heavy_list, heavy_count, nb, seg, nseg, heavy_cap,
                                                giant_list, giant_count, giant_cap, GiantOk<Cv>::value ? giant_threshold() : 0xffffffffu);
    DBG_SYNC("k_fix_heavy");
    if constexpr (GiantOk<Cv>::value) {
        k_giant_partial<KC><<<1024, 256 * LPP, sh_bytes, st>>>(offsets, counts, edge_first, edge_last, giant_list, giant_count, giant_cap, nb, seg, nseg, giant_part);
        k_giant_final<KC><<<64, GIANT_PIECES * LPP, (size_t)GIANT_PIECES * Cv::XW * 4, st>>>(giant_list, giant_count, giant_cap, giant_part, buckets);
        DBG_SYNC("k_giant");
    }
    {   // row sums R_hi (over lo) and column sums C_lo (over hi), level by level in groups of MARG_RADIX; a dimension's last
        // level writes its part of marg (rows first, then columns)
        u32 len_r = RL, len_c = RH;                       // entries still to be summed per row / per column
        const u32* src_r = buckets; const u32* src_c = buckets;
        for (u32 lvl = 0; len_r > 1 || len_c > 1; ++lvl) {
            // groups of MARG_RADIX while there is enough work to fill the chip (throughput-bound levels), pairs afterwards: the
            // small levels are bound by the latency of their dependent additions
            const u32 radix = lvl == 0 || (size_t)RH * len_r >= ((size_t)1 << 17) ? MARG_RADIX : 2;     // (the buffers are sized for a first level of four)
            const u32 rr = len_r > 1 ? (len_r < radix ? len_r : radix) : 0, rc = len_c > 1 ? (len_c < radix ? len_c : radix) : 0;
            const u32 out_len_r = rr ? len_r / rr : 0, out_len_c = rc ? len_c / rc : 0;
            const u32 n_rows = RH * out_len_r, n_cols = out_len_c * RL;
            u32* buf = level_buf[lvl & 1];
            if ((size_t)n_rows + n_cols > (lvl & 1 ? nb / 4 + 2 : nb / 2)) return fail(DVN_ERR_DEVICE, "internal: marginal level %u does not fit its buffer", lvl);
            u32* dst_r = out_len_r == 1 ? marg : buf;
            u32* dst_c = out_len_c == 1 ? marg + (size_t)RH * Cv::XW : buf + (size_t)n_rows * Cv::XW;
            const size_t threads = ((size_t)n_rows + n_cols) * LPP;
            k_marginal_level<KC><<<(unsigned)((threads + MARG_THREADS * LPP - 1) / (MARG_THREADS * LPP)), MARG_THREADS * LPP, 0, st>>>(
                src_r, src_c, lvl == 0, dst_r, dst_c, n_rows, rr, n_cols, rc, kl);
            if (rr) { len_r = out_len_r; src_r = dst_r; }
            if (rc) { len_c = out_len_c; src_c = dst_c; }
        }
    }
    DBG_SYNC("k_marginal_level");
    k_marginal_bitsums<KC><<<npieces, BITSUM_THREADS * LPP, sh_bytes, st>>>(marg, pieces, nb, kl);
    DBG_SYNC("k_marginal_bitsums");
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(L.ev[1], st));
    HIP_TRY(hipMemcpyAsync(L.h_sums, pieces, (size_t)npieces * Cv::XW * 4, hipMemcpyDeviceToHost, st));
    L.red_kh = kh; L.red_kl = kl;
    return DVN_OK;
}

// (re)build the expanded table of an existing set for window size c (0 = back to a plain set)
template <class Cv> static int32_t points_set_expanded(dvn_ctx* ctx, dvn_points* p, u32 c) {
    if (c != 0 && (c < 11 || c > 21)) return fail(DVN_ERR_INVALID_ARG, "window_bits must be 0 or in [11, 21]");
    const u32 nexp = c ? ScalarOf<Cv>::S::BITS / c + 1 : 1;
    if ((size_t)nexp * p->n >= ((size_t)1 << 31)) return fail(DVN_ERR_INVALID_ARG, "expanded table too large");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    u32* nd = nullptr;
    const size_t row = p->n * Cv::AW * 4;
    HIP_TRY(hipMalloc((void**)&nd, row ? row * nexp : 16));
    hipError_t e = row ? hipMemcpyAsync(nd, p->d, row, hipMemcpyDeviceToDevice, ctx->stream) : hipSuccess;
    if (e == hipSuccess && c && p->n) {
        k_points_expand<Cv><<<(unsigned)((p->n + 63) / 64), 64, 0, ctx->stream>>>(nd, p->n, c, nexp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(nd); (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "expansion failed: %s", hipGetErrorString(e)); }
    (void)hipFree(p->d);
    p->d = nd; p->expand_c = c; p->nexp = nexp;
    return DVN_OK;
}

template <class Cv> static typename Cv::X horner(const u32* sums /* nwin * XW internal */, u32 nwin, u32 c) {
    typename Cv::X acc = Cv::infinity();
    for (int w = (int)nwin - 1; w >= 0; --w) {
        for (u32 k = 0; k < c; ++k) acc = Cv::dbl(acc);
        acc = Cv::add(acc, Cv::load_xyzz_internal(sums + (size_t)w * Cv::XW));
    }
    return acc;
}
// affine -> gnark Jacobian image (x, y, 1) / infinity (1, 1, 0)
template <class Cv> static void write_jac(const typename Cv::X& p, uint64_t* out) {
    typedef Ops<decltype(p.x)> O;
    u32* w = (u32*)out;
    const int FW = O::WORDS;
    if (Cv::is_inf(p)) {
        O::to_ext(O::one(), w); O::to_ext(O::one(), w + FW); O::to_ext(O::zero(), w + 2 * FW);
        return;
    }
    typename Cv::A a = Cv::to_affine(p);
    O::to_ext(a.x, w); O::to_ext(a.y, w + FW); O::to_ext(O::one(), w + 2 * FW);
}

// generic finishing arithmetic: the device headers compiled for the host (used where no 64-bit-limb host code exists)
template <class Cv> struct SlowHostG {
    typedef typename Cv::X X;
    static X infinity() { return Cv::infinity(); }
    static bool is_inf(const X& p) { return Cv::is_inf(p); }
    static X add(const X& a, const X& b) { return Cv::add(a, b); }
    static X dbl(const X& a) { return Cv::dbl(a); }
    static X load_device(const u32* w) { return Cv::load_xyzz_internal(w); }
    static X horner(const u32* sums, u32 nwin, u32 c) {
        X acc = Cv::infinity();
        for (int w = (int)nwin - 1; w >= 0; --w) {
            for (u32 k = 0; k < c; ++k) acc = Cv::dbl(acc);
            acc = Cv::add(acc, Cv::load_xyzz_internal(sums + (size_t)w * Cv::XW));
        }
        return acc;
    }
    static void write_jac(const X& p, uint64_t* out) {            // affine normal form with Z = 1, infinity = (1, 1, 0)
        typedef typename Cv::O O;
        u32* w = (u32*)out;
        const int FW = O::WORDS;
        if (Cv::is_inf(p)) { O::to_ext(O::one(), w); O::to_ext(O::one(), w + FW); O::to_ext(O::zero(), w + 2 * FW); return; }
        typename Cv::A a = Cv::to_affine(p);
        O::to_ext(a.x, w); O::to_ext(a.y, w + FW); O::to_ext(O::one(), w + 2 * FW);
    }    // the O(1) finishing arithmetic of a proof (groth16_impl.h): k is a canonical 256-bit integer
    static X mul_scalar(const X& p, const u32 k[8]) { return mul_scalar_n(p, k, 8); }
    static X mul_scalar_n(const X& p, const u32* k, u32 nwords) {
        X acc = Cv::infinity();
        for (int b = (int)nwords * 32 - 1; b >= 0; --b) {
            acc = Cv::dbl(acc);
            if ((k[b >> 5] >> (b & 31)) & 1) acc = Cv::add(acc, p);
        }
        return acc;
    }
    static X load_device_affine(const u32* w) { return Cv::from_affine(Cv::load_affine_internal(w)); }
    static X normalize(const X& p) { return Cv::from_affine(Cv::to_affine(p)); }
};
template <class Cv> struct FastHost { typedef SlowHostG<Cv> G; };
template <> struct FastHost<G1> { typedef dvn_host::HG1 G; };
template <> struct FastHost<G2> { typedef dvn_host::HG2 G; };
#ifndef DVN_SLOW_HOST            // -DDVN_SLOW_HOST: the device headers compiled for the host everywhere but BN254 (A/B checks)
template <> struct FastHost<G1_377> { typedef dvn_host::Group<dvn_host::OpsFqN<Fp377Cfg, 6>> G; };
template <> struct FastHost<G2_377> { typedef dvn_host::Group<dvn_host::OpsFq2N<Fp377Cfg, 6, 5>> G; };
template <> struct FastHost<G1_381> { typedef dvn_host::Group<dvn_host::OpsFqN<Fp381Cfg, 6>> G; };
template <> struct FastHost<G2_381> { typedef dvn_host::Group<dvn_host::OpsFq2N<Fp381Cfg, 6, 1>> G; };
template <> struct FastHost<G_761> { typedef dvn_host::Group<dvn_host::OpsFqN<Fp761Cfg, 12>> G; };
#endif

// ---- lanes: submit (enqueue on a free lane) / wait --------------------------------------------------------------
static inline int32_t lane_acquire(dvn_ctx* ctx, MsmLane** out) {
    for (int k = 0; k < DVN_LANES; ++k) {
        MsmLane& L = ctx->lane[k];
        if (!L.busy) {
            // work queued earlier on the context's main stream (gathers, computeH, uploads) must be visible to this lane
            // (a caller that has queued main-stream work the lane does NOT depend on records ev_dep itself and sets dep_hold)
            if (!ctx->dep_hold) HIP_TRY(hipEventRecord(ctx->ev_dep, ctx->stream));
            HIP_TRY(hipStreamWaitEvent(L.stream, ctx->ev_dep, 0));
            L.busy = true; L.seq++;
            L.ext_ws = nullptr; L.ext_bytes = 0;
            *out = &L;
            return DVN_OK;
        }
    }
    return fail(DVN_ERR_INVALID_ARG, "too many MSMs in flight (collect one of the %d outstanding tickets first)", DVN_LANES);
}
static inline int32_t lane_wait(dvn_ctx* ctx, MsmLane& L) {
    hipError_t e = hipStreamSynchronize(L.stream);
    L.busy = false;
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "MSM failed on the device: %s", hipGetErrorString(e)); }
    HIP_TRY(hipEventElapsedTime(&L.ms[0], L.ev[0], L.ev[1]));
    HIP_TRY(hipEventElapsedTime(&L.ms[1], L.ev[2], L.ev[3]));
    ctx->last_ms[0] = L.ms[0]; ctx->last_ms[1] = L.ms[1];
    return DVN_OK;
}
// queue a full MSM (all windows) on a free lane; *lane_out identifies it
template <class Cv>
static int32_t msm_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, MsmLane** lane_out,
                          bool static_scratch = false) {
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "len(points) != len(scalars): range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    if (static_scratch && !ctx->static_ws) return fail(DVN_ERR_DEVICE, "static scratch buffer was not resolved at dvn_init");
    MsmLane* L = nullptr;
    TRY(lane_acquire(ctx, &L));
    if (static_scratch) { L->ext_ws = ctx->static_ws; L->ext_bytes = ctx->static_ws_bytes; }
    L->curve = pts->curve; L->suite = pts->suite; L->W = 0;
    int32_t rc = DVN_OK;
    if (n == 0) { L->nwin = 0; }
    else if (use_expanded(pts, n)) {
        L->expanded = true; L->c = 0; L->nwin = 1;
        rc = msm_expanded<Cv>(ctx, *L, pts, off, n, scalars, flags, 0, pts->nexp);
    } else {
        u32 c, nwin; msm_plan(n, &c, &nwin, ScalarOf<Cv>::S::BITS);
        L->expanded = false; L->c = c; L->nwin = nwin;
        rc = msm_windows<Cv>(ctx, *L, pts, off, n, scalars, flags, c, nwin, 0, nwin);
    }
    if (rc != DVN_OK) { (void)hipStreamSynchronize(L->stream); L->busy = false; return rc; }
    *lane_out = L;
    return DVN_OK;
}
// Share `rank` of `world` of one MSM (device scalars): ONE partial result; the partial results of all ranks add up to the MSM
// (this is what one GPU of a sharded prover computes).  An expanded set (row w = 2^(c w) P) is shared by ENTRY COUNT: its
// nexp * n (row, point) entries, in row-major order, are cut into `world` equal ranges -- 13 rows over 8 GPUs are 1.625 rows
// each, not 2 / 2 / 2 / 2 / 2 / 1 / 1 / 1 -- a rank runs the rows its range touches with the entries outside the range
// masked to zero digits.  A set too small to be expanded (< 2^10 points: microseconds of work) is computed whole by rank 0,
// the other ranks contribute the point at infinity.
template <class Cv>
static int32_t msm_share_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, u32 rank, u32 world, MsmLane** lane_out,
                                uint32_t flags = DVN_MSM_SCALARS_ON_DEVICE) {
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "len(points) != len(scalars): range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    if (world == 0 || rank >= world) return fail(DVN_ERR_INVALID_ARG, "rank %u outside world %u", rank, world);
    if (n && !use_expanded(pts, n)) {
        if (rank == 0) return msm_submit<Cv>(ctx, pts, off, n, (const uint64_t*)d_scalars, flags, lane_out);
        n = 0;
    }
    MsmLane* L = nullptr;
    TRY(lane_acquire(ctx, &L));
    L->curve = pts->curve; L->suite = pts->suite; L->W = 0; L->expanded = true; L->c = 0;
    const u64 flat_total = (u64)pts->nexp * n;
    const u64 flat_lo = flat_total * rank / world, flat_hi = flat_total * (rank + 1) / world;
    L->nwin = (n && flat_hi > flat_lo) ? 1 : 0;
    int32_t rc = DVN_OK;
    if (L->nwin) {
        const u32 row_lo = (u32)(flat_lo / n), row_hi = (u32)((flat_hi + n - 1) / n);
        rc = msm_expanded<Cv>(ctx, *L, pts, off, n, (const uint64_t*)d_scalars, flags, row_lo, row_hi, flat_lo, flat_hi);
    }
    if (rc != DVN_OK) { (void)hipStreamSynchronize(L->stream); L->busy = false; return rc; }
    *lane_out = L;
    return DVN_OK;
}
// B scalar vectors (device, vector b at [b * n, (b + 1) * n)) over the same n points -> B internal xyzz images in d_out
// (device).  ENQUEUE only, on a free lane; lane_wait() completes it.  Used by the blob KZG cell proofs (dvn_kzg.hip): 128
// quotient polynomials committed over one monomial basis.
template <class Cv>
static int32_t msm_batched_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, u32 batch, u32 c, u32* d_out, MsmLane** lane_out,
                                  const u32* row_table = nullptr) {
    if (off > pts->n || n > pts->n - off) return fail(DVN_ERR_INVALID_ARG, "len(points) != len(scalars): range [%zu, %zu) outside point set of %zu", off, off + n, pts->n);
    if (n == 0 || batch == 0 || c < 2 || c > 16) return fail(DVN_ERR_INVALID_ARG, "batched MSM: empty input or window size outside [2, 16]");
    MsmLane* L = nullptr;
    TRY(lane_acquire(ctx, &L));
    const u32 nwin = ScalarOf<Cv>::S::BITS / c + 1;
    L->curve = pts->curve; L->suite = pts->suite; L->W = 0; L->expanded = false; L->c = c; L->nwin = nwin;
    int32_t rc = msm_windows_staged<Cv>(ctx, *L, pts, off, n, (const uint64_t*)d_scalars, DVN_MSM_SCALARS_ON_DEVICE, c, 0, nwin, batch, d_out, row_table);
    if (rc != DVN_OK) { (void)hipStreamSynchronize(L->stream); L->busy = false; return rc; }
    *lane_out = L;
    return DVN_OK;
}

// expanded table for the batched path: `rows` rows of the first n points of pts (row w = 2^(c w) * P_i); caller frees it
template <class Cv>
static int32_t msm_row_table_create(dvn_ctx* ctx, const dvn_points* pts, size_t n, u32 c, u32* rows_out, u32** table_out) {
    if (n == 0 || n > pts->n) return fail(DVN_ERR_INVALID_ARG, "row table: bad point count");
    const u32 rows = ScalarOf<Cv>::S::BITS / c + 1;
    u32* t = nullptr;
    HIP_TRY(hipMalloc((void**)&t, (size_t)rows * n * Cv::AW * 4));
    hipError_t e = hipMemcpyAsync(t, pts->d, n * Cv::AW * 4, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) { k_points_expand<Cv><<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(t, n, c, rows); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(t); (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "row table: %s", hipGetErrorString(e)); }
    *rows_out = rows; *table_out = t;
    return DVN_OK;
}
// complete it: host-side window combine (host_fast.h)
template <class Cv>
static int32_t msm_collect(dvn_ctx* ctx, MsmLane& L, typename FastHost<Cv>::G::X* out) {
    typedef typename FastHost<Cv>::G HG;
    if (L.nwin == 0) { L.busy = false; *out = HG::infinity(); ctx->last_ms[0] = ctx->last_ms[1] = 0.f; return DVN_OK; }
    TRY(lane_wait(ctx, L));
    if (L.expanded) *out = bitsum_combine<HG>(L.h_sums, L.red_kh, L.red_kl, Cv::XW);
    else *out = HG::horner(L.h_sums, L.nwin, L.c);
    return DVN_OK;
}

// device scalars -> host (4x64-limb) extended-Jacobian point (caller holds the lock)
template <class Cv>
static int32_t msm_dev_xyzz(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, typename FastHost<Cv>::G::X* out) {
    MsmLane* L = nullptr;
    TRY(msm_submit<Cv>(ctx, pts, off, n, (const uint64_t*)d_scalars, DVN_MSM_SCALARS_ON_DEVICE, &L));
    return msm_collect<Cv>(ctx, *L, out);
}

template <class Cv>
static int32_t msm_full(dvn_ctx* ctx, const dvn_points* pts, int curve, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, uint64_t* out_jac) {
    TRY(enter(ctx));
    if (!pts || !out_jac || (!scalars && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->curve != curve || pts->suite != SuiteOf<Cv>::value) return fail(DVN_ERR_INVALID_ARG, "point set is G%d of suite %d, call is G%d of suite %d", pts->curve, pts->suite, curve, SuiteOf<Cv>::value);
    typedef typename FastHost<Cv>::G HG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane* L = nullptr;
    TRY(msm_submit<Cv>(ctx, pts, off, n, scalars, flags, &L));
    typename HG::X r;
    TRY(msm_collect<Cv>(ctx, *L, &r));
    HG::write_jac(r, out_jac);      // MSB-first window combine happened on the host (host_fast.h)
    return DVN_OK;
}
// asynchronous pair: submit returns a ticket (lane index + 1), collect finishes that MSM
template <class Cv>
static int32_t msm_submit_api(dvn_ctx* ctx, const dvn_points* pts, int curve, size_t off, size_t n, const uint64_t* scalars, uint32_t flags, int32_t* ticket) {
    TRY(enter(ctx));
    if (!pts || !ticket || (!scalars && n)) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pts->curve != curve || pts->suite != SuiteOf<Cv>::value) return fail(DVN_ERR_INVALID_ARG, "point set is G%d of suite %d, call is G%d of suite %d", pts->curve, pts->suite, curve, SuiteOf<Cv>::value);
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane* L = nullptr;
    TRY(msm_submit<Cv>(ctx, pts, off, n, scalars, flags, &L));
    *ticket = (int32_t)(L - ctx->lane) + 1;
    return DVN_OK;
}
template <class Cv>
static int32_t msm_collect_api(dvn_ctx* ctx, int curve, int32_t ticket, uint64_t* out_jac) {
    TRY(enter(ctx));
    if (!out_jac || ticket < 1 || ticket > DVN_LANES) return fail(DVN_ERR_INVALID_ARG, "bad ticket");
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmLane& L = ctx->lane[ticket - 1];
    if (!L.busy || L.curve != curve || L.suite != SuiteOf<Cv>::value) return fail(DVN_ERR_INVALID_ARG, "ticket %d is not an outstanding G%d MSM of this curve", ticket, curve);
    typedef typename FastHost<Cv>::G HG;
    typename HG::X r;
    TRY(msm_collect<Cv>(ctx, L, &r));
    HG::write_jac(r, out_jac);
    return DVN_OK;
}
