// Pippenger MSM kernels for BN254 G1 / G2 on gfx950.
//
// Replaces gnark-crypto's G1Jac.MultiExp / G2Jac.MultiExp, which groth16.Prove calls
// five times per proof (/root/reference/prover/prover_cpu.go:37 -> gnark backend/groth16/bn254/prove.go).
//
// Pipeline (DESIGN.md "MSM"):
//   k_digits16    scalar (Montgomery) -> canonical -> signed c-bit digits (int16) for every window
//   k_tile_hist / k_tile_offsets / k_scan / k_tile_scatter
//                 per-window counting sort of (point index | sign) by bucket: 64K-entry tiles, the whole 2^(c-1)-
//                 counter histogram of a tile lives in LDS (128 KiB at c = 16), every rank comes from an LDS atomic
//   k_accumulate  HOT KERNEL: one lane per fixed-size SEGMENT of the sorted list (not per bucket), so every
//                 lane of a wave does exactly SEG mixed additions whatever the scalar distribution is; runs
//                 that a segment shares with its neighbours are written as edge partials
//   k_fix / k_fix_heavy   stitch the edge partials of buckets that span segments (heavy = block-cooperative)
//   k_bucket_reduce + k_window_finalize   sum_b b * B_b per window: chunked running sums + small scalar mul +
//                 LDS tree
//   host          Horner combine of the <= 32 window sums (O(254) group operations)
#pragma once
#include "bn254_curve.h"

namespace dvn {

constexpr u32 SIGN_BIT = 0x80000000u;

__device__ __forceinline__ void load_words16(const u32* __restrict__ p, u32* w, int nwords) {
    // nwords is a multiple of 4 and p is 16-byte aligned
    for (int i = 0; i < nwords; i += 4) {
        uint4 v = *reinterpret_cast<const uint4*>(p + i);
        w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
    }
}
__device__ __forceinline__ void store_words16(u32* __restrict__ p, const u32* w, int nwords) {
    for (int i = 0; i < nwords; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(w[i], w[i + 1], w[i + 2], w[i + 3]);
}

template <class Cv> __device__ __forceinline__ typename Cv::A ld_affine(const u32* __restrict__ p) {
    u32 w[Cv::AW];
#pragma unroll
    for (int i = 0; i < Cv::AW; i += 4) {
        uint4 v = *reinterpret_cast<const uint4*>(p + i);
        w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
    }
    return Cv::load_affine_internal(w);
}
template <class Cv> __device__ __forceinline__ typename Cv::X ld_xyzz(const u32* __restrict__ p) {
    u32 w[Cv::XW];
#pragma unroll
    for (int i = 0; i < Cv::XW; i += 4) {
        uint4 v = *reinterpret_cast<const uint4*>(p + i);
        w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
    }
    return Cv::load_xyzz_internal(w);
}
template <class Cv> __device__ __forceinline__ void st_xyzz(const typename Cv::X& pt, u32* __restrict__ p) {
    u32 w[Cv::XW];
    Cv::store_xyzz_internal(pt, w);
#pragma unroll
    for (int i = 0; i < Cv::XW; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(w[i], w[i + 1], w[i + 2], w[i + 3]);
}

// ---- points: external gnark image -> internal image, in place ------------------------------------------
template <class Cv> __global__ void k_points_to_internal(u32* __restrict__ pts, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[Cv::AW];
    load_words16(pts + i * Cv::AW, w, Cv::AW);
    typename Cv::A p = Cv::affine_from_ext(w);
    Cv::store_affine_internal(p, w);
    store_words16(pts + i * Cv::AW, w, Cv::AW);
}
template <class Cv> __global__ void k_points_to_external(const u32* __restrict__ pts, u32* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename Cv::A p = ld_affine<Cv>(pts + i * Cv::AW);
    u32 w[Cv::AW];
    Cv::affine_to_ext(p, w);
    store_words16(out + i * Cv::AW, w, Cv::AW);
}
// P_i = base + i*step, written in internal affine form; base/step are internal affine images
template <class Cv> __global__ void k_points_progression(u32* __restrict__ pts, size_t n, const u32* __restrict__ base_step, u32 chunk) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * chunk;
    if (lo >= n) return;
    size_t hi = lo + chunk < n ? lo + chunk : n;
    typename Cv::A base = ld_affine<Cv>(base_step), step = ld_affine<Cv>(base_step + Cv::AW);
    typename Cv::X cur = Cv::add_mixed(Cv::mul_small(Cv::from_affine(step), (u32)lo), base);
    for (size_t i = lo; i < hi; ++i) {
        typename Cv::A a = Cv::to_affine(cur);
        u32 w[Cv::AW];
        Cv::store_affine_internal(a, w);
        store_words16(pts + i * Cv::AW, w, Cv::AW);
        cur = Cv::add_mixed(cur, step);
    }
}

// ---- synthetic scalar streams (bench/test inputs; oracle/bn254_py.py fr_at / witness_like_scalar) -----------
__device__ __forceinline__ u64 dev_splitmix_at(u64 seed, u64 index) {
    u64 z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// kind 0: uniform mod r; kind 1: 40 % zero, 25 % one, 25 % < 2^64, 10 % uniform.  Output gnark Montgomery words.
template <class S> __global__ void k_scalars_synthetic(u32* __restrict__ out, size_t n, u64 seed, int kind) {
    constexpr int EW = S::EW, QW = S::EW / 2;          // external u32 words, 64-bit stream words per scalar
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[EW];
#pragma unroll
    for (int j = 0; j < EW; ++j) w[j] = 0;
    int mode = 3;   // 0 zero, 1 one, 2 one word, 3 full
    if (kind == 1) {
        const u64 sel = dev_splitmix_at(seed ^ 0x5EED5EED5EED5EEDULL, i) % 100;
        mode = sel < 40 ? 0 : sel < 65 ? 1 : sel < 90 ? 2 : 3;
    }
    if (mode == 1) w[0] = 1;
    else if (mode == 2) { u64 v = dev_splitmix_at(seed, QW * i); w[0] = (u32)v; w[1] = (u32)(v >> 32); }
    else if (mode == 3) {
        for (int k = 0; k < QW; ++k) { u64 v = dev_splitmix_at(seed, QW * i + k); w[2 * k] = (u32)v; w[2 * k + 1] = (u32)(v >> 32); }
        // reduce the 32*EW-bit value mod the field order: repeated conditional subtraction
        u32 rm[EW];
        { u32 t[EW]; Fe<S> pm;
#pragma unroll
          for (int j = 0; j < S::NL; ++j) pm.v[j] = S::p(j);
          fe_pack<S>(pm, t);
#pragma unroll
          for (int j = 0; j < EW; ++j) rm[j] = t[j]; }
        for (int rep = 0; rep < 256; ++rep) {
            bool ge = true;
            for (int j = EW - 1; j >= 0; --j) { if (w[j] > rm[j]) { ge = true; break; } if (w[j] < rm[j]) { ge = false; break; } }
            if (!ge) break;
            u64 br = 0;
            for (int j = 0; j < EW; ++j) { u64 d = (u64)w[j] - rm[j] - br; w[j] = (u32)d; br = (d >> 32) & 1; }
        }
    }
    u32 o[EW];
    fe_to_ext_mont<S>(fe_from_canonical<S>(fe_unpack<S>(w)), o);
    store_words16(out + i * EW, o, EW);
}

// ---- digits -> per-window counting sort, all ranking done with LDS atomics (no global atomics) ------------------
// Signed c-bit digits in [-2^(c-1), 2^(c-1) - 1] stored as int16 (c <= 16): digits[(w - win_lo) * n + i].
// Bucket index of a non-zero digit d is |d| - 1 in [0, 2^(c-1)).
template <class S> __global__ void k_digits16(const u32* __restrict__ scalars, size_t n, u32 c, u32 win_lo, u32 win_hi,
                                  int16_t* __restrict__ digits) {
    constexpr int EW = S::EW;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[EW];
    load_words16(scalars + i * EW, w, EW);
    Fe<S> k = fe_ext_mont_to_canonical<S>(w);
    u32 kw[EW + 1];
    {
        u32 t[EW]; fe_pack<S>(k, t);
#pragma unroll
        for (int j = 0; j < EW; ++j) kw[j] = t[j];
        kw[EW] = 0;
    }
    const u32 nb = 1u << (c - 1), mask = (1u << c) - 1;
    u32 carry = 0;
    for (u32 win = 0; win < win_hi; ++win) {
        const u32 bit = win * c;
        u32 d = carry;
        if (bit < 32 * EW) {
            const u32 wi = bit >> 5, sh = bit & 31;
            u64 two = (u64)kw[wi] | ((u64)kw[wi + 1] << 32);
            d += (u32)(two >> sh) & mask;
        }
        carry = 0;
        int v = (int)d;
        if (d >= nb) { v = (int)d - (int)(1u << c); carry = 1; }
        if (win >= win_lo) digits[(size_t)(win - win_lo) * n + i] = (int16_t)v;
    }
}

// block (w, tile): histogram of the tile's bucket indices in LDS -> hist[(w * ntiles + tile) * nb + b]
static __global__ __launch_bounds__(1024) void k_tile_hist(const int16_t* __restrict__ digits, size_t n, u32 nb, u32 tile_len,
                                                           u32 ntiles, u32* __restrict__ hist) {
    extern __shared__ __align__(16) u32 lh[];
    const u32 w = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)tile * tile_len, hi = min(lo + (size_t)tile_len, n);
    const int16_t* dg = digits + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int d = dg[i];
        if (d) atomicAdd(&lh[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    u32* out = hist + (size_t)blockIdx.x * nb;
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) out[b] = lh[b];
}
// thread (w, b): exclusive prefix of hist over the tiles of window w, in place; counts[w][b] = total
static __global__ void k_tile_offsets(u32* __restrict__ hist, u32* __restrict__ counts, u32 nb, u32 ntiles, u32 nwin_local) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)nwin_local * nb) return;
    const u32 w = (u32)(g / nb), b = (u32)(g % nb);
    u32 run = 0;
    u32* p = hist + (size_t)w * ntiles * nb + b;
    for (u32 t = 0; t < ntiles; ++t, p += nb) { const u32 v = *p; *p = run; run += v; }
    counts[g] = run;
}
// block (w, tile): final position of every entry = offsets[w][b] + hist[w][tile][b] + rank inside the tile
static __global__ __launch_bounds__(1024) void k_tile_scatter(const int16_t* __restrict__ digits, size_t n, u32 nb, u32 tile_len,
                                                              u32 ntiles, const u32* __restrict__ hist,
                                                              const u32* __restrict__ offsets, u32* __restrict__ sorted) {
    extern __shared__ __align__(16) u32 lh[];
    const u32 w = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    const u32* hb = hist + (size_t)blockIdx.x * nb;
    const u32* ob = offsets + (size_t)w * nb;
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = hb[b] + ob[b];
    __syncthreads();
    const size_t lo = (size_t)tile * tile_len, hi = min(lo + (size_t)tile_len, n);
    const int16_t* dg = digits + (size_t)w * n;
    u32* dst = sorted + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int d = dg[i];
        if (d) {
            const u32 pos = atomicAdd(&lh[(d < 0 ? -d : d) - 1], 1u);
            dst[pos] = (u32)i | (d < 0 ? SIGN_BIT : 0u);
        }
    }
}

// one block per window: offsets = exclusive scan of counts; totals[w] = number of entries
static __global__ void k_scan(const u32* __restrict__ counts, u32* __restrict__ offsets, u32* __restrict__ totals, u32 nb) {
    __shared__ u32 part[1024];
    const u32 w = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    const u32 per = (nb + T - 1) / T;
    const u32 lo = t * per, hi = min(lo + per, nb);
    const u32* cw = counts + (size_t)w * nb;
    u32 s = 0;
    for (u32 b = lo; b < hi; ++b) s += cw[b];
    part[t] = s;
    __syncthreads();
    for (u32 d = 1; d < T; d <<= 1) {
        u32 v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - s;
    for (u32 b = lo; b < hi; ++b) { offsets[(size_t)w * nb + b] = run; run += cw[b]; }
    if (t == T - 1) totals[w] = part[t];
}

// ====================================================================================================================
// Expanded point sets ("one bucket set").  HBM is 288 GB, a proving key a few GB: at upload the table
// is widened to rows  row_w[i] = 2^(c w) * P_i,  w < nexp.  Every window of every scalar then lands in the SAME
// 2^(c-1) buckets, so window size is no longer tied to the bucket-reduction cost: c = 20 gives 13 windows instead of
// 16 (19 % fewer mixed additions), one bucket reduction instead of 16 and no window combine at all.
// Sorting 2^19 buckets no longer fits an LDS histogram: two passes -- by the high bits of the bucket (LDS histogram of
// <= 2048 bins per 64K-entry tile), then one block per high-bits bin sorts its contiguous segment by the low 10 bits.
// ====================================================================================================================
constexpr u32 LOB = 10;                       // low bucket bits sorted in pass 2 (2^LOB threads per pass-2 block)
template <class Cv> __global__ void k_points_expand(u32* __restrict__ table, size_t n, u32 c, u32 nexp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename Cv::A a = ld_affine<Cv>(table + i * Cv::AW);
    for (u32 w = 1; w < nexp; ++w) {
        typename Cv::X cur = Cv::from_affine(a);
        for (u32 k = 0; k < c; ++k) cur = Cv::dbl(cur);
        a = Cv::to_affine(cur);
        u32 wds[Cv::AW];
        Cv::store_affine_internal(a, wds);
        store_words16(table + ((size_t)w * n + i) * Cv::AW, wds, Cv::AW);
    }
}
// signed c-bit digits (c <= 24) as int32: digits[(w - win_lo) * n + i] in [-2^(c-1), 2^(c-1) - 1]
template <class S> __global__ void k_digits32(const u32* __restrict__ scalars, size_t n, u32 c, u32 win_lo, u32 win_hi, int* __restrict__ digits) {
    constexpr int EW = S::EW;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[EW];
    load_words16(scalars + i * EW, w, EW);
    Fe<S> k = fe_ext_mont_to_canonical<S>(w);
    u32 kw[EW + 1];
    {
        u32 t[EW]; fe_pack<S>(k, t);
#pragma unroll
        for (int j = 0; j < EW; ++j) kw[j] = t[j];
        kw[EW] = 0;
    }
    const u32 nb = 1u << (c - 1), mask = (1u << c) - 1;
    u32 carry = 0;
    for (u32 win = 0; win < win_hi; ++win) {
        const u32 bit = win * c;
        u32 d = carry;
        if (bit < 32 * EW) {
            const u32 wi = bit >> 5, sh = bit & 31;
            u64 two = (u64)kw[wi] | ((u64)kw[wi + 1] << 32);
            d += (u32)(two >> sh) & mask;
        }
        carry = 0;
        int v = (int)d;
        if (d >= nb) { v = (int)d - (int)(1u << c); carry = 1; }
        if (win >= win_lo) digits[(size_t)(win - win_lo) * n + i] = v;
    }
}
// pass 1, block = tile of the flattened (window, point) entry space: histogram over the high bucket bits
static __global__ __launch_bounds__(1024) void k_p1_hist(const int* __restrict__ digits, size_t total, u32 tile_len, u32 hb,
                                                         u32* __restrict__ hist1) {
    extern __shared__ __align__(16) u32 lh[];
    for (u32 b = threadIdx.x; b < hb; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)blockIdx.x * tile_len, hi = min(lo + (size_t)tile_len, total);
    for (size_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        const int d = digits[g];
        if (d) atomicAdd(&lh[((u32)(d < 0 ? -d : d) - 1) >> LOB], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < hb; b += blockDim.x) hist1[(size_t)blockIdx.x * hb + b] = lh[b];
}
// block = high bin: exclusive prefix of hist1 over the tiles (in place), bin_total[bin]
static __global__ __launch_bounds__(1024) void k_p1_offsets(u32* __restrict__ hist1, u32* __restrict__ bin_total, u32 hb, u32 ntiles) {
    __shared__ u32 part[1024];
    const u32 bin = blockIdx.x, t = threadIdx.x, T = blockDim.x;
    const u32 per = (ntiles + T - 1) / T;
    const u32 lo = t * per, hi = min(lo + per, ntiles);
    u32 s = 0;
    for (u32 k = lo; k < hi; ++k) s += hist1[(size_t)k * hb + bin];
    part[t] = s;
    __syncthreads();
    for (u32 d = 1; d < T; d <<= 1) {
        u32 v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - s;
    for (u32 k = lo; k < hi; ++k) { const u32 v = hist1[(size_t)k * hb + bin]; hist1[(size_t)k * hb + bin] = run; run += v; }
    if (t == T - 1) bin_total[bin] = part[t];
}
// single block: bin_base = exclusive scan of bin_total (hb <= 2048), bin_base[hb] = totals[0] = number of entries
static __global__ __launch_bounds__(1024) void k_p1_scan(const u32* __restrict__ bin_total, u32* __restrict__ bin_base,
                                                         u32* __restrict__ totals, u32 hb) {
    __shared__ u32 part[1024];
    const u32 t = threadIdx.x, T = blockDim.x;
    const u32 per = (hb + T - 1) / T;
    const u32 lo = t * per, hi = min(lo + per, hb);
    u32 s = 0;
    for (u32 k = lo; k < hi; ++k) s += bin_total[k];
    part[t] = s;
    __syncthreads();
    for (u32 d = 1; d < T; d <<= 1) {
        u32 v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - s;
    for (u32 k = lo; k < hi; ++k) { bin_base[k] = run; run += bin_total[k]; }
    if (t == T - 1) { bin_base[hb] = part[t]; totals[0] = part[t]; }
}
// pass 1 scatter: entry (w, i) -> position inside its high bin; carries ref = row * stride + point | sign and the low bits
static __global__ __launch_bounds__(1024) void k_p1_scatter(const int* __restrict__ digits, size_t total, size_t n, u32 tile_len,
                                                            u32 hb, u32 win_lo, u32 row_stride, u32 pts_off,
                                                            const u32* __restrict__ hist1, const u32* __restrict__ bin_base,
                                                            u32* __restrict__ p1_ref, unsigned short* __restrict__ p1_lo) {
    extern __shared__ __align__(16) u32 lh[];
    for (u32 b = threadIdx.x; b < hb; b += blockDim.x) lh[b] = bin_base[b] + hist1[(size_t)blockIdx.x * hb + b];
    __syncthreads();
    const size_t lo = (size_t)blockIdx.x * tile_len, hi = min(lo + (size_t)tile_len, total);
    for (size_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        const int d = digits[g];
        if (d) {
            const u32 bucket = (u32)(d < 0 ? -d : d) - 1;
            const u32 w = (u32)(g / n), i = (u32)(g - (size_t)w * n);
            const u32 pos = atomicAdd(&lh[bucket >> LOB], 1u);
            p1_ref[pos] = ((win_lo + w) * row_stride + pts_off + i) | (d < 0 ? SIGN_BIT : 0u);
            p1_lo[pos] = (unsigned short)(bucket & ((1u << LOB) - 1));
        }
    }
}
// pass 2, block = high bin: counting sort of the bin's contiguous segment by the low bits; emits the per-bucket
// counts / offsets the accumulate kernel needs
static __global__ __launch_bounds__(1024) void k_p2_sort(const u32* __restrict__ bin_base, const u32* __restrict__ p1_ref,
                                                         const unsigned short* __restrict__ p1_lo, u32* __restrict__ sorted,
                                                         u32* __restrict__ counts, u32* __restrict__ offsets) {
    __shared__ u32 cnt[1u << LOB];
    __shared__ u32 cur[1u << LOB];
    const u32 bin = blockIdx.x, t = threadIdx.x;           // blockDim.x == 1 << LOB
    const u32 s0 = bin_base[bin], s1 = bin_base[bin + 1];
    cnt[t] = 0;
    __syncthreads();
    for (u32 e = s0 + t; e < s1; e += blockDim.x) atomicAdd(&cnt[p1_lo[e]], 1u);
    __syncthreads();
    const u32 mine = cnt[t];
    cur[t] = mine;
    __syncthreads();
    for (u32 d = 1; d < blockDim.x; d <<= 1) {
        u32 v = (t >= d) ? cur[t - d] : 0;
        __syncthreads();
        cur[t] += v;
        __syncthreads();
    }
    const u32 off = s0 + cur[t] - mine;
    __syncthreads();
    cur[t] = off;
    counts[((size_t)bin << LOB) + t] = mine;
    offsets[((size_t)bin << LOB) + t] = off;
    __syncthreads();
    for (u32 e = s0 + t; e < s1; e += blockDim.x) {
        const u32 pos = atomicAdd(&cur[p1_lo[e]], 1u);
        sorted[pos] = p1_ref[e];
    }
}

// ---- HOT KERNEL: segment accumulation -------------------------------------------------------------------
// lane = (window w, segment k): entries [k*seg, min((k+1)*seg, totals[w])) of window w's sorted list.
template <class Cv>
__global__ __launch_bounds__(128) void k_accumulate(const u32* __restrict__ points, size_t pts_off,
                                                    const u32* __restrict__ sorted, const u32* __restrict__ offsets,
                                                    const u32* __restrict__ counts, const u32* __restrict__ totals,
                                                    u32* __restrict__ bucket_sums, u32* __restrict__ edge_first,
                                                    u32* __restrict__ edge_last, size_t n, u32 nb, u32 seg, u32 nseg,
                                                    u32 nwin_local) {
    typedef typename Cv::X X;
    typedef typename Cv::A A;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 w = (u32)(g / nseg);
    if (w >= nwin_local) return;
    const u32 k = (u32)(g - (size_t)w * nseg);
    const u32 total = totals[w];
    const u32 pos0 = k * seg;
    if (pos0 >= total) return;
    const u32 pos1 = min(pos0 + seg, total);
    const u32* off = offsets + (size_t)w * nb;
    const u32* cnt = counts + (size_t)w * nb;
    const u32* src = sorted + (size_t)w * n;
    // bucket containing pos0: largest b with off[b] <= pos0 (the last of equal offsets is the non-empty one)
    u32 lo = 0, hi = nb;
    while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (off[mid] <= pos0) lo = mid; else hi = mid; }
    u32 b = lo;
    u32 b_start = off[b], b_end = b_start + cnt[b];
    X acc = Cv::infinity();
    // Flat loop: every lane of the wave executes exactly (pos1 - pos0) additions; a run boundary is a short
    // predicated flush inside the iteration, never a separate inner loop (which would make the wave iterate
    // max-run-length times per run and double the instruction count).
    auto flush = [&](u32 end_x) {
        u32* dst;
        if (b_start >= pos0 && b_end <= pos1) dst = bucket_sums + ((size_t)w * nb + b) * Cv::XW;
        else if (b_start <= pos0) dst = edge_first + ((size_t)w * nseg + k) * Cv::XW;
        else dst = edge_last + ((size_t)w * nseg + k) * Cv::XW;
        st_xyzz<Cv>(Cv::acc_normalize(acc), dst);
        (void)end_x;
    };
    for (u32 x = pos0; x < pos1; ++x) {
        if (x == b_end) {                       // bucket b is exhausted: store its run, move to the next non-empty bucket
            flush(x);
            acc = Cv::infinity();
            do { ++b; } while (cnt[b] == 0);
            b_start = off[b]; b_end = b_start + cnt[b];
        }
        const u32 e = src[x];
        A p = ld_affine<Cv>(points + (pts_off + (size_t)(e & ~SIGN_BIT)) * Cv::AW);
        if (e & SIGN_BIT) p = Cv::neg(p);
        acc = Cv::acc_add_mixed(acc, p);
    }
    flush(pos1);
}

// lane per bucket: empty -> infinity; spanning <= FIX_LIGHT segments -> stitch serially; more -> heavy list
constexpr u32 FIX_LIGHT = 24;
template <class Cv>
__global__ __launch_bounds__(128) void k_fix(const u32* __restrict__ offsets, const u32* __restrict__ counts,
                                             u32* __restrict__ bucket_sums, const u32* __restrict__ edge_first,
                                             const u32* __restrict__ edge_last, u32* __restrict__ heavy_list,
                                             u32* __restrict__ heavy_count, u32 nwin_local, u32 nb, u32 seg, u32 nseg,
                                             u32 heavy_cap) {
    typedef typename Cv::X X;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)nwin_local * nb) return;
    const u32 w = (u32)(g / nb);
    const u32 c = counts[g];
    u32* dst = bucket_sums + g * Cv::XW;
    if (c == 0) { st_xyzz<Cv>(Cv::infinity(), dst); return; }
    const u32 o = offsets[g], e = o + c;
    const u32 ks = o / seg, ke = (e - 1) / seg;
    if (ks == ke) return;                       // complete inside one segment: written by k_accumulate
    if (ke - ks + 1 > FIX_LIGHT) {
        const u32 slot = atomicAdd(heavy_count, 1u);
        if (slot < heavy_cap) heavy_list[slot] = (u32)g;
        return;
    }
    const u32* first = edge_first + (size_t)w * nseg * Cv::XW;
    const u32* last = edge_last + (size_t)w * nseg * Cv::XW;
    X acc = ld_xyzz<Cv>(((o % seg) == 0 ? first : last) + (size_t)ks * Cv::XW);
    for (u32 k = ks + 1; k <= ke; ++k) acc = Cv::add(acc, ld_xyzz<Cv>(first + (size_t)k * Cv::XW));
    st_xyzz<Cv>(acc, dst);
}

// block-cooperative sum of points in shared memory (blockDim.x power of two <= 256); result in lane 0
template <class Cv> __device__ typename Cv::X block_sum(typename Cv::X v, u32* sh /* blockDim.x * XW words */) {
    const u32 t = threadIdx.x;
    for (u32 stride = blockDim.x >> 1; stride >= 1; stride >>= 1) {
        if (t >= stride && t < 2 * stride) Cv::store_xyzz_internal(v, sh + (size_t)(t - stride) * Cv::XW);
        __syncthreads();
        if (t < stride) v = Cv::add(v, Cv::load_xyzz_internal(sh + (size_t)t * Cv::XW));
        __syncthreads();
    }
    return v;
}

template <class Cv>
__global__ __launch_bounds__(256) void k_fix_heavy(const u32* __restrict__ offsets, const u32* __restrict__ counts,
                                                   u32* __restrict__ bucket_sums, const u32* __restrict__ edge_first,
                                                   const u32* __restrict__ edge_last, const u32* __restrict__ heavy_list,
                                                   const u32* __restrict__ heavy_count, u32 nb, u32 seg, u32 nseg, u32 heavy_cap) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 nheavy = min(*heavy_count, heavy_cap);
    for (u32 h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const u32 g = heavy_list[h];
        const u32 w = g / nb;
        const u32 o = offsets[g], e = o + counts[g];
        const u32 ks = o / seg, ke = (e - 1) / seg;
        const u32* first = edge_first + (size_t)w * nseg * Cv::XW;
        const u32* last = edge_last + (size_t)w * nseg * Cv::XW;
        X acc = Cv::infinity();
        for (u32 k = ks + threadIdx.x; k <= ke; k += blockDim.x) {
            const u32* src = (k == ks && (o % seg) != 0) ? last : first;
            acc = Cv::add(acc, ld_xyzz<Cv>(src + (size_t)k * Cv::XW));
        }
        acc = block_sum<Cv>(acc, sh);
        if (threadIdx.x == 0) st_xyzz<Cv>(acc, bucket_sums + (size_t)g * Cv::XW);
        __syncthreads();
    }
}

// ---- bucket reduction: window sum = sum_{b=1..nb} b * B_b -------------------------------------------------
// block handles blockDim.x * L consecutive buckets of one window; thread handles L of them.
template <class Cv>
__global__ __launch_bounds__(256) void k_bucket_reduce(const u32* __restrict__ bucket_sums, u32* __restrict__ partials,
                                                       u32 nb, u32 L, u32 blocks_per_window) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 w = blockIdx.x / blocks_per_window, blk = blockIdx.x % blocks_per_window;
    const u32 first = (blk * blockDim.x + threadIdx.x) * L;     // 0-based index of this thread's first bucket
    X run = Cv::infinity(), tot = Cv::infinity();
    if (first < nb) {
        const u32 last = min(first + L, nb);
        const u32* base = bucket_sums + (size_t)w * nb * Cv::XW;
        for (u32 b = last; b-- > first;) {
            run = Cv::add(run, ld_xyzz<Cv>(base + (size_t)b * Cv::XW));
            tot = Cv::add(tot, run);
        }
        // tot = sum_j (j - first + 1) * B_j ; bucket j (0-based) has weight j + 1 = (j - first + 1) + first
        if (first) tot = Cv::add(tot, Cv::mul_small(run, first));
    }
    tot = block_sum<Cv>(tot, sh);
    if (threadIdx.x == 0) st_xyzz<Cv>(tot, partials + (size_t)blockIdx.x * Cv::XW);
}
// one small block per window: tree-sum of the window's partial results
template <class Cv>
__global__ __launch_bounds__(64) void k_window_finalize(const u32* __restrict__ partials, u32* __restrict__ window_sums, u32 nwin_local,
                                                        u32 blocks_per_window) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 w = blockIdx.x;
    X acc = Cv::infinity();
    for (u32 i = threadIdx.x; i < blocks_per_window; i += blockDim.x)
        acc = Cv::add(acc, ld_xyzz<Cv>(partials + ((size_t)w * blocks_per_window + i) * Cv::XW));
    acc = block_sum<Cv>(acc, sh);
    if (threadIdx.x == 0) st_xyzz<Cv>(acc, window_sums + (size_t)w * Cv::XW);
    (void)nwin_local;
}

}  // namespace dvn
