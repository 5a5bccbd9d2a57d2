// Host drivers of fft.Domain / FFT / FFTInverse, templated on the scalar-field config S (FrCfg for BN254, Fr377Cfg for
// BLS12-377); included by dvn_ntt.hip and dvn_bls12377.hip.
#pragma once
#include "dvn_internal.h"
#include "ntt_kernels.h"

// per-field constants of gnark-crypto's fft.Domain: 2-adic root of unity (canonical words), its order, the coset shift
template <class S> struct DomainConsts;
template <> struct DomainConsts<FrCfg> {      // w_{2^28} = 5^((r-1)/2^28), FrMultiplicativeGen = 5
    static constexpr u32 ADICITY = 28; static constexpr u64 COSET_GEN = 5; static constexpr int SUITE = 0;
    static const u64* root() { static const u64 r[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL}; return r; }
};
template <> struct DomainConsts<Fr377Cfg> {   // w_{2^47} = 22^((r-1)/2^47) = gnark-crypto's 8065159656716812877374967518403273466521432693661810619979959746626482506078
    static constexpr u32 ADICITY = 47; static constexpr u64 COSET_GEN = 22; static constexpr int SUITE = 1;
    static const u64* root() { static const u64 r[4] = {0x476ef4a4ec2a895eULL, 0x9b506ee363e3f04aULL, 0x60c69477d1a8a12fULL, 0x11d4b7f60cb92cc1ULL}; return r; }
};
template <> struct DomainConsts<Fr381Cfg> {   // w_{2^32} = 7^((r-1)/2^32) = 10238227357739495823651030575849232062558860180284477541189508159991286009131
    static constexpr u32 ADICITY = 32; static constexpr u64 COSET_GEN = 7; static constexpr int SUITE = 3;
    static const u64* root() { static const u64 r[4] = {0x3829971f439f0d2bULL, 0xb63683508c2280b9ULL, 0xd09b681922c813b4ULL, 0x16a2a19edfe81f20ULL}; return r; }
};
template <> struct DomainConsts<Fp377Cfg> {   // BW6-761's scalar field: w_{2^46} = 15^((r-1)/2^46) = gnark-crypto's
    // 32863578547254505029601261939868325669770508939375122462904745766352256812585773382134936404344547323199885654433
    static constexpr u32 ADICITY = 46; static constexpr u64 COSET_GEN = 15; static constexpr int SUITE = 2;
    static const u64* root() { static const u64 r[6] = {0x7eca603cc563b9a1ULL, 0x06df0a4306fe0bc3ULL, 0xb44d994a0ddff8c6ULL, 0x40fbe05b4512a3d4ULL, 0x30f152488aeffc9bULL, 0x0036a92e05198a80ULL}; return r; }
};

// ------------------------------------------------------------------------------------------------------
// NTT
// ------------------------------------------------------------------------------------------------------
template <class S> static Fe<S> fr_from_u64(u64 v) {
    u32 w[S::EW] = {(u32)v, (u32)(v >> 32)};
    return fe_from_canonical<S>(fe_unpack<S>(w));
}
template <class S> static Fe<S> fr_root_of_unity(u32 log_n) {      // square the 2-adic root down to order 2^log_n
    Fe<S> w = fe_from_canonical<S>(fe_unpack<S>(*(const u32(*)[S::EW])DomainConsts<S>::root()));
    for (u32 i = log_n; i < DomainConsts<S>::ADICITY; ++i) w = fe_sqr<S>(w);
    return w;
}
template <class S> static void fr_store_internal(const Fe<S>& v, u32* w) { u32 t[S::EW]; fe_pack<S>(fe_canonical<S>(v), t); memcpy(w, t, S::EW * 4); }

template <class S> static int32_t launch_powers(dvn_ctx* ctx, u32* d_out, const Fe<S>& base, const Fe<S>& first, size_t count, u32* d_scratch16) {
    u32 h[2 * S::EW]; fr_store_internal<S>(base, h); fr_store_internal<S>(first, h + S::EW);
    HIP_TRY(hipMemcpyAsync(d_scratch16, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    const u32 chunk = 64;
    size_t threads = (count + chunk - 1) / chunk;
    k_powers<S><<<(unsigned)((threads + 63) / 64), 64, 0, ctx->stream>>>(d_out, d_scratch16, count, chunk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // h goes out of scope
    return DVN_OK;
}
// entries [e0, e0 + count) of a twiddle table of n_entries entries in limb planes: first * base^k
template <class S> static int32_t launch_powers_planes(dvn_ctx* ctx, u32* d_table, u32 n_entries, u32 e0, const Fe<S>& base, const Fe<S>& first, size_t count, u32* d_scratch16) {
    u32 h[2 * S::EW]; fr_store_internal<S>(base, h); fr_store_internal<S>(first, h + S::EW);
    HIP_TRY(hipMemcpyAsync(d_scratch16, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    const u32 chunk = 64;
    size_t threads = (count + chunk - 1) / chunk;
    k_powers_planes<S><<<(unsigned)((threads + 63) / 64), 64, 0, ctx->stream>>>(d_table, n_entries, e0, d_scratch16, count, chunk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DVN_OK;
}

static inline void ntt_domain_release(dvn_domain* d) {            // no locking, no ctx: direct frees (caller synchronised)
    if (!d) return;
    u32* ptrs[] = {d->tw_fwd, d->tw_inv, d->tw_fwd_c, d->tw_inv_c, d->g_lo, d->g_hi, d->gi_lo, d->gi_hi, d->card_inv, d->h_consts};
    for (u32* p : ptrs) if (p) (void)hipFree(p);
    delete d;
}

template <class S> static int32_t ntt_domain_create(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out) {
    TRY(enter(ctx));
    if (!out) return fail(DVN_ERR_INVALID_ARG, "out is NULL");
    if (log_n < 1 || log_n > 27 || log_n > DomainConsts<S>::ADICITY) return fail(DVN_ERR_INVALID_ARG, "log_n = %u outside [1, 27]", log_n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    dvn_domain* d = new (std::nothrow) dvn_domain();
    if (!d) return fail(DVN_ERR_OOM, "host allocation failed");
    d->log_n = log_n; d->suite = DomainConsts<S>::SUITE;
    const size_t n = (size_t)1 << log_n;
    d->lo_bits = log_n < 10 ? log_n : 10;
    const size_t nlo = (size_t)1 << d->lo_bits, nhi = n >> d->lo_bits;
    u32* scratch = nullptr;
    int32_t rc = DVN_OK;
    auto alloc = [&](u32** p, size_t entries) -> int32_t { HIP_TRY(hipMalloc((void**)p, entries * S::EW * 4)); return DVN_OK; };
    // a twiddle table: n entries of NL limbs in planes (entry n - 1 is never written or read)
    auto alloc_tw = [&](u32** p) -> int32_t { HIP_TRY(hipMalloc((void**)p, n * S::NL * 4)); HIP_TRY(hipMemsetAsync(*p, 0, n * S::NL * 4, ctx->stream)); return DVN_OK; };
    do {
        if ((rc = alloc_tw(&d->tw_fwd)) || (rc = alloc_tw(&d->tw_inv)) || (rc = alloc_tw(&d->tw_fwd_c)) || (rc = alloc_tw(&d->tw_inv_c)) ||
            (rc = alloc(&d->g_lo, nlo)) || (rc = alloc(&d->g_hi, nhi)) ||
            (rc = alloc(&d->gi_lo, nlo)) || (rc = alloc(&d->gi_hi, nhi)) || (rc = alloc(&d->card_inv, 1)) || (rc = alloc(&d->h_consts, 2)) ||
            (rc = alloc(&scratch, 2))) break;
        const Fe<S> one = fe_one<S>();
        const Fe<S> w = fr_root_of_unity<S>(log_n), wi = fe_inv<S>(w);
        const Fe<S> g = fr_from_u64<S>(DomainConsts<S>::COSET_GEN), gi = fe_inv<S>(g), ninv = fe_inv<S>(fr_from_u64<S>(n));
        Fe<S> ws = w, wis = wi, gs = g, gis = gi;               // w^(2^s), w^-(2^s), g^(2^s), g^-(2^s)
        for (u32 s = 0; s < log_n && rc == DVN_OK; ++s) {
            const size_t half = n >> (s + 1);
            const u32 offs = (u32)(n - (n >> s));
            if ((rc = launch_powers_planes<S>(ctx, d->tw_fwd, (u32)n, offs, ws, one, half, scratch))) break;
            if ((rc = launch_powers_planes<S>(ctx, d->tw_inv, (u32)n, offs, wis, one, half, scratch))) break;
            if ((rc = launch_powers_planes<S>(ctx, d->tw_fwd_c, (u32)n, offs, ws, gs, half, scratch))) break;
            if ((rc = launch_powers_planes<S>(ctx, d->tw_inv_c, (u32)n, offs, wis, gis, half, scratch))) break;
            ws = fe_sqr<S>(ws); wis = fe_sqr<S>(wis); gs = fe_sqr<S>(gs); gis = fe_sqr<S>(gis);
        }
        if (rc) break;
        if ((rc = launch_powers<S>(ctx, d->g_lo, g, one, nlo, scratch))) break;
        if ((rc = launch_powers<S>(ctx, d->g_hi, fe_pow_u64<S>(g, nlo), one, nhi, scratch))) break;
        if ((rc = launch_powers<S>(ctx, d->gi_lo, gi, ninv, nlo, scratch))) break;
        if ((rc = launch_powers<S>(ctx, d->gi_hi, fe_pow_u64<S>(gi, nlo), one, nhi, scratch))) break;
        constexpr u32 EW = S::EW;
        u32 h[3 * EW];
        fr_store_internal<S>(ninv, h);
        // computeH runs its seven transforms without any 1/n: the pointwise constants carry them.  With A', B', C' the unscaled
        // coset evaluations (n times the true ones) and the last inverse unscaled as well, h = raw[A' B' den / n^3 - C' den / n^2].
        // k1 = den * 2^EXT_SHIFT / n^3 (the product of two external words is read 2^-EXT_SHIFT too small), k2 = den / n^2
        const Fe<S> den = fe_inv<S>(fe_sub<S>(fe_pow_u64<S>(g, n), one));
        const Fe<S> ninv2 = fe_sqr<S>(ninv);
        fr_store_internal<S>(fe_mul<S>(fe_mul<S>(den, fr_from_u64<S>(1u << S::EXT_SHIFT)), fe_mul<S>(ninv2, ninv)), h + EW);
        fr_store_internal<S>(fe_mul<S>(den, ninv2), h + 2 * EW);
        hipError_t e = hipMemcpy(d->card_inv, h, EW * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d->h_consts, h + EW, 2 * EW * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = fail(DVN_ERR_DEVICE, "domain constant upload failed: %s", hipGetErrorString(e)); }
    } while (0);
    if (scratch) (void)hipFree(scratch);
    if (rc != DVN_OK) {
        std::string keep = tl_error;
        ntt_domain_release(d);
        tl_error = keep;
        return rc;
    }
    *out = d;
    return DVN_OK;
}

constexpr size_t NTT_TILE_LDS_MAX = 73728;
// dynamic LDS above the 64 KiB default needs the attribute once per kernel and device (with that device current)
template <class S> static int32_t ntt_attrs(int device) {
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64) return fail(DVN_ERR_INVALID_ARG, "device ordinal %d out of range", device);
    if (!done[device]) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_ntt_pass<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_TILE_LDS_MAX));
        done[device] = true;
    }
    return DVN_OK;
}
// what computeH asks of a transform beyond the public modes
struct NttOpt {
    bool no_ninv = false;                              // inverse transforms: leave the 1/n out
    const u32* hb = nullptr; const u32* hc = nullptr;  // first pass loads data[i] * hb[i] * k1 - hc[i] * k2 (k1, k2 = the domain's h_consts)
};
// passes of up to NTT_KMAX stages: tiles of 2^K sub-problem elements times I neighbours (>= 64-byte runs in HBM)
constexpr u32 NTT_KMAX = 10;
// in-place transform of device data; asynchronous on ctx->stream
template <class S> static int32_t ntt_device(dvn_ctx* ctx, const dvn_domain* d, u32* data, int inverse, int decimation, int coset, const NttOpt& opt = NttOpt()) {
    RoctxRange rr(inverse ? "dvn:ntt_inverse" : "dvn:ntt_forward");
    const u32 log_n = d->log_n;
    static const u32 kmax = [] { const char* e = getenv("DVN_NTT_KMAX"); const int v = e ? atoi(e) : 0; return (u32)(v >= 2 && v <= 11 ? v : NTT_KMAX); }();
    // stage counts per pass in execution order (host_fast.h: the pass with the transform's last two stages gets an even count)
    u32 ks[16];
    const u32 npass = dvn_host::ntt_pass_plan(log_n, kmax, decimation == DVN_DIF, ks);
    // the butterfly network absorbs the coset scaling where its shape allows (DIT: a scaling of the input, DIF: of the output)
    const bool merged = coset && ((!inverse && decimation == DVN_DIT) || (inverse && decimation == DVN_DIF));
    const u32* tw = inverse ? (merged ? d->tw_inv_c : d->tw_inv) : (merged ? d->tw_fwd_c : d->tw_fwd);
    NttScale none{nullptr, nullptr, 0, 0, 0, 0, nullptr};
    NttScale sc = none;
    if (!inverse && coset && !merged) sc = NttScale{d->g_lo, d->g_hi, d->lo_bits, 2, 0, 1, nullptr};                 // DIF: g^i at load
    if (inverse && (!coset || merged) && !opt.no_ninv) sc = NttScale{d->card_inv, nullptr, 0, 1, 0, 2, nullptr};     // 1/n at store
    if (inverse && coset && !merged) sc = NttScale{d->gi_lo, d->gi_hi, d->lo_bits, 2, 0, 2, nullptr};                // DIT: g^-i / n at store
    NttScale pointwise{opt.hb, opt.hc, 0, 3, 0, 1, d->h_consts};
    if (opt.hb && (sc.mode && sc.when == 1)) return fail(DVN_ERR_INVALID_ARG, "internal: pointwise load on a pre-scaled transform");
    // stage ranges in execution order
    for (u32 p = 0; p < npass; ++p) {
        u32 sa, sb;
        if (decimation == DVN_DIF) { sa = 0; for (u32 q = 0; q < p; ++q) sa += ks[q]; sb = sa + ks[p]; }
        else { sb = log_n; for (u32 q = 0; q < p; ++q) sb -= ks[q]; sa = sb - ks[p]; }
        const u32 K = sb - sa;
        u32 log_i = 3;
        while (K + log_i > log_n) --log_i;                 // tile must not exceed the array
        // an LDS slot is the element's NL limbs (ntt_kernels.h): tiles of 2048 nine-limb (72 KiB) / 1024 fourteen-limb (56 KiB)
        // elements, two blocks per CU
        while (log_i > 0 && ((size_t)1 << (K + log_i)) * S::NL * 4 > NTT_TILE_LDS_MAX) --log_i;
        const size_t lds = ((size_t)1 << (K + log_i)) * S::NL * 4;
        if (lds > NTT_TILE_LDS_MAX) return fail(DVN_ERR_DEVICE, "internal: a pass of %u stages does not fit the LDS tile", K);
        const u32 tiles = (u32)(((size_t)1 << log_n) >> (K + log_i));
        TRY(ntt_attrs<S>(ctx->device));
        NttScale use = none;
        if (sc.mode && ((sc.when == 1 && p == 0) || (sc.when == 2 && p == npass - 1))) use = sc;
        if (opt.hb && p == 0) use = pointwise;
        const int triv = !merged && sb == log_n && K >= 2;
        k_ntt_pass<S><<<tiles, 512, lds, ctx->stream>>>(data, tw, log_n, sa, sb, decimation == DVN_DIT, p == npass - 1, log_i, triv, use);
        HIP_TRY(hipGetLastError());
    }
    return DVN_OK;
}

// computeH core on device buffers (gnark prove.go computeH): 3 x FFTInverse(DIF), 3 x FFT(DIT, coset),
// pointwise (a*b - c) * den, FFTInverse(DIF, coset).  Result in da, bit-reversed; db and dc are left holding intermediate
// values.  No transform scales: the 1/n factors live in the pointwise constants, the coset shifts in the twiddles, and the
// pointwise step itself runs in the load of the last transform.  Caller holds ctx->mu.
template <class S> static int32_t compute_h_device(dvn_ctx* ctx, const dvn_domain* dom, u32* da, u32* db, u32* dc) {
    u32* v[3] = {da, db, dc};
    NttOpt raw; raw.no_ninv = true;
    for (int i = 0; i < 3; ++i) TRY(ntt_device<S>(ctx, dom, v[i], 1, DVN_DIF, 0, raw));
    for (int i = 0; i < 3; ++i) TRY(ntt_device<S>(ctx, dom, v[i], 0, DVN_DIT, 1));
    NttOpt fin; fin.no_ninv = true; fin.hb = db; fin.hc = dc;
    TRY(ntt_device<S>(ctx, dom, da, 1, DVN_DIF, 1, fin));
    return DVN_OK;
}
// the C-ABI entry (host or device a, b, c of `len` <= n elements, zero-padded): shared by every scalar field
template <class S> static int32_t compute_h_api(dvn_ctx* ctx, const char* name, const dvn_domain* dom, const uint64_t* a, const uint64_t* b,
                                                const uint64_t* c, size_t len, uint64_t* h_out, uint32_t flags) {
    TRY(enter(ctx));
    if (!dom || !a || !b || !c || !h_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (dom->suite != DomainConsts<S>::SUITE) return fail(DVN_ERR_INVALID_ARG, "domain is not a %s domain", name);
    const size_t n = (size_t)1 << dom->log_n;
    if (len > n) return fail(DVN_ERR_INVALID_ARG, "len = %zu exceeds domain size %zu", len, n);
    std::lock_guard<std::mutex> lk(ctx->mu);
    const bool on_dev = flags & DVN_NTT_DATA_ON_DEVICE;
    const hipMemcpyKind in_kind = on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    constexpr size_t EB = S::EW * 4;                               // bytes per element
    TRY(ensure_ws(ctx, 3 * n * EB + 1024));
    u32* da = (u32*)ctx->ws; u32* db = da + n * S::EW; u32* dc = db + n * S::EW;
    hipStream_t st = ctx->stream;
    const uint64_t* in[3] = {a, b, c}; u32* dv[3] = {da, db, dc};
    for (int q = 0; q < 3; ++q) {
        if (on_dev) HIP_TRY(hipMemcpyAsync(dv[q], in[q], len * EB, in_kind, st));
        else TRY(dvn_h2d(ctx, dv[q], in[q], len * EB, st));
    }
    for (u32* v : {da, db, dc}) TRY(dvn_zero_async(v + len * S::EW, (n - len) * EB, st));          // zero padding to the domain size
    HIP_TRY(hipEventRecord(ctx->ev[0], st));
    TRY(compute_h_device<S>(ctx, dom, da, db, dc));
    HIP_TRY(hipEventRecord(ctx->ev[1], st));
    HIP_TRY(hipMemcpyAsync(h_out, da, n * EB, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&ctx->last_ms[0], ctx->ev[0], ctx->ev[1]));
    ctx->last_ms[1] = ctx->last_ms[0];
    return DVN_OK;
}
