// Pippenger MSM kernels for BN254 G1 / G2 on gfx950.
//
// Replaces gnark-crypto's G1Jac.MultiExp / G2Jac.MultiExp, which groth16.Prove calls
// five times per proof (/root/reference/prover/prover_cpu.go:37 -> gnark backend/groth16/bn254/prove.go).
//
// Pipeline:
//   k_digits16/32 scalar (Montgomery) -> canonical -> signed c-bit digits for every window
//   k_s1_* / k_s2_*   two-pass LDS-staged counting sort of (point ref | sign) by (window, bucket): both passes rank
//                 with LDS atomics, reorder their 8192-entry tile in LDS and write runs
//   k_accumulate  HOT KERNEL: one lane per fixed-size SEGMENT of the sorted list (not per bucket), so every
//                 lane of a wave does exactly SEG mixed additions whatever the scalar distribution is; runs
//                 that a segment shares with its neighbours are written as edge partials
//   k_fix / k_fix_heavy / k_giant_*   stitch the edge partials of buckets that span segments (heavy = block-
//                 cooperative, giant = 64 blocks per bucket)
//   k_bucket_reduce + k_window_finalize   sum_b b * B_b per window: chunked running sums + small scalar mul +
//                 LDS tree
//   host          Horner combine of the <= 32 window sums (O(254) group operations)
#pragma once
#include "bn254_curve.h"
#include "host_fast.h"
#include "fp2_paired.h"

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

// Lane layout of a kernel's curve type: 1 lane per point, or 2 for the paired Fp2 arithmetic (fp2_paired.h: one Fp2
// component per lane).  Kernels index their work by LOGICAL thread: ltid / lbdim / lgtid.
template <class Cv> struct Lanes { static constexpr u32 LPP = 1; };
template <class C> struct Lanes<Curve<Fe2P<C>>> { static constexpr u32 LPP = 2; };
template <class Cv> __device__ __forceinline__ u32 ltid() { return threadIdx.x / Lanes<Cv>::LPP; }
template <class Cv> __device__ __forceinline__ u32 lbdim() { return blockDim.x / Lanes<Cv>::LPP; }
template <class Cv> __device__ __forceinline__ size_t lgtid() { return ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / Lanes<Cv>::LPP; }

template <class Cv> __device__ __forceinline__ typename Cv::A ld_affine(const u32* __restrict__ p) {
    if constexpr (Lanes<Cv>::LPP == 2) return ld_affine_paired<Cv>(p);
    else {
        u32 w[Cv::AW];
#pragma unroll
        for (int i = 0; i < Cv::AW; i += 4) {
            uint4 v = *reinterpret_cast<const uint4*>(p + i);
            w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
        }
        return Cv::load_affine_internal(w);
    }
}
template <class Cv> __device__ __forceinline__ typename Cv::X ld_xyzz(const u32* __restrict__ p) {
    if constexpr (Lanes<Cv>::LPP == 2) return ld_xyzz_paired<Cv>(p);
    else {
        u32 w[Cv::XW];
#pragma unroll
        for (int i = 0; i < Cv::XW; i += 4) {
            uint4 v = *reinterpret_cast<const uint4*>(p + i);
            w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
        }
        return Cv::load_xyzz_internal(w);
    }
}
template <class Cv> __device__ __forceinline__ void st_xyzz(const typename Cv::X& pt, u32* __restrict__ p) {
    if constexpr (Lanes<Cv>::LPP == 2) st_xyzz_paired<Cv>(pt, p);
    else {
        u32 w[Cv::XW];
        Cv::store_xyzz_internal(pt, w);
#pragma unroll
        for (int i = 0; i < Cv::XW; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(w[i], w[i + 1], w[i + 2], w[i + 3]);
    }
}

// ---- raw accumulator images: what k_accumulate hands to the fix / reduction kernels --------------------------------------
// A run's sum leaves the hot loop as its 29-bit limbs, one u32 each (4 coordinates x NL words; the paired Fp2 curves: the even
// lane's 4 NL words, then the odd lane's), NOT as the packed canonical image of st_xyzz.  A flush then is a handful of stores:
// no reduction of the loose x coordinate, no canonicalisation, no bit packing -- in the hot loop that work ran for the whole
// wave whenever any one of its 64 lanes ended a run (about every second iteration at 2^22 points), on the consumers' side it
// runs once per lane.  Infinity is an all-zero ZZ, as in the packed image.
template <class Cv> struct AccImg {
    static constexpr int NL = Cv::O::Cfg::NL;
    static constexpr int LANE_W = 4 * NL;                             // words one lane moves
    static constexpr int W = LANE_W * (int)Lanes<Cv>::LPP;            // words per image
    static_assert(LANE_W % 4 == 0, "16-byte accesses");
};
template <class C> __device__ __forceinline__ Fe<C>& fe_limbs(Fe<C>& a) { return a; }
template <class C> __device__ __forceinline__ Fe<C>& fe_limbs(Fe2P<C>& a) { return a.c; }
template <class C> __device__ __forceinline__ const Fe<C>& fe_limbs(const Fe<C>& a) { return a; }
template <class C> __device__ __forceinline__ const Fe<C>& fe_limbs(const Fe2P<C>& a) { return a.c; }
template <class Cv> __device__ __forceinline__ void st_acc(const typename Cv::X& pt, u32* __restrict__ p) {
    constexpr int NL = AccImg<Cv>::NL, LW = AccImg<Cv>::LANE_W;
    u32 w[LW];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        w[i] = fe_limbs(pt.x).v[i]; w[NL + i] = fe_limbs(pt.y).v[i];
        w[2 * NL + i] = fe_limbs(pt.zz).v[i]; w[3 * NL + i] = fe_limbs(pt.zzz).v[i];
    }
    u32* q = p;
    if constexpr (Lanes<Cv>::LPP == 2) q += pair_odd() * LW;
#pragma unroll
    for (int i = 0; i < LW; i += 4) *reinterpret_cast<uint4*>(q + i) = make_uint4(w[i], w[i + 1], w[i + 2], w[i + 3]);
}
template <class Cv> __device__ __forceinline__ typename Cv::X ld_acc(const u32* __restrict__ p) {
    constexpr int NL = AccImg<Cv>::NL, LW = AccImg<Cv>::LANE_W;
    typedef typename Cv::O::Cfg C;
    const u32* q = p;
    if constexpr (Lanes<Cv>::LPP == 2) q += pair_odd() * LW;
    u32 w[LW];
#pragma unroll
    for (int i = 0; i < LW; i += 4) { const uint4 v = *reinterpret_cast<const uint4*>(q + i); w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w; }
    typename Cv::X pt;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        fe_limbs(pt.x).v[i] = w[i]; fe_limbs(pt.y).v[i] = w[NL + i];
        fe_limbs(pt.zz).v[i] = w[2 * NL + i]; fe_limbs(pt.zzz).v[i] = w[3 * NL + i];
    }
    if (Cv::is_inf(pt)) return Cv::infinity();
    // the lazily reduced accumulator leaves x below 8p: back to the weak form the general group law stores and compares
    if constexpr (Cv::O::LAZY) fe_limbs(pt.x) = fe_reduce_full<C>(fe_limbs(pt.x));
    return pt;
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

// Batched layout (B scalar vectors over ONE point set, msm_driver.h msm_windows_staged(batch > 1)): scalar s = b * n + i,
// every (vector, window) pair is its own virtual window: digits[((b * nwin + win) * n + i], nwin = all windows of a scalar.
template <class S> __global__ void k_digits16_batched(const u32* __restrict__ scalars, size_t n, u32 batch, u32 c, u32 nwin,
                                                      int16_t* __restrict__ digits) {
    constexpr int EW = S::EW;
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n * batch) return;
    const size_t b = s / n, i = s % n;
    u32 w[EW];
    load_words16(scalars + s * EW, w, EW);
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
    for (u32 win = 0; win < nwin; ++win) {
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
        digits[((size_t)b * nwin + win) * n + i] = (int16_t)v;
    }
}

// ====================================================================================================================
// Expanded point sets ("one bucket set").  HBM is 288 GB, a proving key a few GB: at upload the table
// is widened to rows  row_w[i] = 2^(c w) * P_i,  w < nexp.  Every window of every scalar then lands in the SAME
// 2^(c-1) buckets, so window size is no longer tied to the bucket-reduction cost: c = 20 gives 13 windows instead of
// 16 (19 % fewer mixed additions), one bucket reduction instead of 16 and no window combine at all.
// Sorting 2^19 buckets no longer fits an LDS histogram: the two-pass staged sort below splits the bucket bits evenly.
// ====================================================================================================================
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
// ====================================================================================================================
// Two-pass LDS-staged sort: the scattered 4-byte stores of a one-pass counting sort leave L2 as one
// request each (2^26 requests per 2^22-point MSM, ~1.3 ms), so both passes here reorder their tile in LDS first and
// write RUNS: consecutive lanes store consecutive addresses.
//   pass 1  tile = ST consecutive points of one window: bin = bucket >> lob (lb = nb >> lob bins per key-window),
//           LDS histogram -> ranks, LDS reorder by bin, coalesced copy-out of (ref, low bits) into the bin's segment
//   pass 2  block = one bin: counts its 2^lob sub-buckets (emits the per-bucket counts/offsets the accumulate kernel
//           needs), then sorts the segment chunk by chunk through LDS the same way.
// Keys: key = kw * nb + bucket, kw = the window for plain sets ("keyed"), 0 for expanded sets (one bucket set).
// ====================================================================================================================
constexpr u32 ST = 8192;                       // entries per pass-1 tile and per pass-2 chunk (8 per thread)
// exclusive prefix of v over the block (1024 threads); *total = block sum.  wsum: >= 80 words of LDS.  The caller
// separates two calls by a __syncthreads().
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* wsum, u32* total) {
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d, 64); if (lane >= (u32)d) inc += o; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        const u32 sv = lane < (blockDim.x >> 6) ? wsum[lane] : 0;
        u32 si = sv;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(si, d, 64); if (lane >= (u32)d) si += o; }
        wsum[lane] = si - sv;
        if (lane == 63) wsum[64] = si;
    }
    __syncthreads();
    *total = wsum[64];
    return wsum[wid] + inc - v;
}
template <class D> __device__ __forceinline__ u32 digit_bucket(D d) { const int v = (int)d; return (u32)(v < 0 ? -v : v) - 1; }

// pass 1a, block = group (w, g) of tpb consecutive tiles of window w (gpw groups per window): histogram of the group over its
// key-window's lb bins.  One histogram row per GROUP, not per tile: the row array (and the prefix pass over it) shrinks by tpb.
template <class D> __global__ __launch_bounds__(1024) void k_s1_hist(const D* __restrict__ digits, size_t n, u32 gpw, u32 tpb, u32 lob, u32 lb,
                                                                   u32* __restrict__ hist1) {
    extern __shared__ __align__(16) u32 sm[];
    const u32 w = blockIdx.x / gpw, g = blockIdx.x % gpw;
    for (u32 b = threadIdx.x; b < lb; b += blockDim.x) sm[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)g * tpb * ST;
    const u32 cnt = (u32)min((size_t)tpb * ST, n - lo);
    const D* dg = digits + (size_t)w * n + lo;
    for (u32 e = threadIdx.x; e < cnt; e += blockDim.x) {
        const D d = dg[e];
        if (d) atomicAdd(&sm[digit_bucket(d) >> lob], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < lb; b += blockDim.x) hist1[(size_t)blockIdx.x * lb + b] = sm[b];
}
// Expanded sets: signed c-bit digits (c <= 24) as int32, digits[(w - win_lo) * n + i] in [-2^(c-1), 2^(c-1) - 1].
// flat_lo / flat_hi: only the entries whose flat position win * n + i lies in [flat_lo, flat_hi) are kept, the others become
// zero digits (no entry): an MSM over an expanded table is shared between GPUs by entry count, not by whole rows.
// Digit extraction with pass 1a of the sort folded in: block g takes tpb * ST consecutive scalars, writes ALL their
// windows' digits and keeps one LDS histogram per window, so the digit array is not read back for the histogram.
// hist1 row of (window w, group g) = (w * gpw + g), as k_s1_hist writes it.  LDS: W * lb words.
template <class S> __global__ __launch_bounds__(1024) void k_digits32_hist(const u32* __restrict__ scalars, size_t n, u32 c, u32 win_lo, u32 win_hi,
                                                                         int* __restrict__ digits, u64 flat_lo, u64 flat_hi,
                                                                         u32 gpw, u32 tpb, u32 lob, u32 lb, u32* __restrict__ hist1) {
    constexpr int EW = S::EW;
    extern __shared__ __align__(16) u32 sm[];
    const u32 W = win_hi - win_lo, g = blockIdx.x, tid = threadIdx.x;
    for (u32 j = tid; j < W * lb; j += blockDim.x) sm[j] = 0;
    __syncthreads();
    const size_t lo = (size_t)g * tpb * ST, hi = min(n, lo + (size_t)tpb * ST);
    const u32 nb = 1u << (c - 1), mask = (1u << c) - 1;
    for (size_t i = lo + tid; i < hi; i += blockDim.x) {
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
            if (win >= win_lo) {
                const u64 flat = (u64)win * n + i;
                if (!(flat >= flat_lo && flat < flat_hi)) v = 0;
                digits[(size_t)(win - win_lo) * n + i] = v;
                if (v) atomicAdd(&sm[(win - win_lo) * lb + (digit_bucket(v) >> lob)], 1u);
            }
        }
    }
    __syncthreads();
    for (u32 j = tid; j < W * lb; j += blockDim.x) hist1[((size_t)(j / lb) * gpw + g) * lb + (j % lb)] = sm[j];
}
// pass 1b: exclusive prefix of hist1 over the tk rows (tile groups) of every key-window, per bin, in two coalesced steps
// (a thread owns a bin, so consecutive lanes read consecutive words of a row):
//   k_s1_offsets_a  block (kw, j) walks rows [j * S1_RPB, (j + 1) * S1_RPB) of key-window kw: every row is replaced by the
//                   prefix inside the block, the block's column sums go to part[(kw * nblk + j) * lb + bin]
//   k_s1_offsets_b  block kw turns part[] into exclusive prefixes over j and writes bin_total[kw * lb + bin]
// The position of (row r, bin) inside its bin's segment is then hist1[r][bin] + part[r / S1_RPB][bin] (k_s1_scatter adds them).
constexpr u32 S1_RPB = 64;
static __global__ __launch_bounds__(1024) void k_s1_offsets_a(u32* __restrict__ hist1, u32* __restrict__ part, u32 lb, u32 tk, u32 nblk) {
    const u32 kw = blockIdx.x / nblk, j = blockIdx.x % nblk;
    const u32 r0 = j * S1_RPB, r1 = min(tk, r0 + S1_RPB);
    for (u32 bin = threadIdx.x; bin < lb; bin += blockDim.x) {
        u32* h = hist1 + ((size_t)kw * tk + r0) * lb + bin;
        u32 run = 0;
        for (u32 r = r0; r < r1; ++r, h += lb) { const u32 v = *h; *h = run; run += v; }
        part[(size_t)blockIdx.x * lb + bin] = run;
    }
}
static __global__ __launch_bounds__(1024) void k_s1_offsets_b(u32* __restrict__ part, u32* __restrict__ bin_total, u32 lb, u32 nblk) {
    const u32 kw = blockIdx.x;
    for (u32 bin = threadIdx.x; bin < lb; bin += blockDim.x) {
        u32* q = part + (size_t)kw * nblk * lb + bin;
        u32 run = 0;
        for (u32 j = 0; j < nblk; ++j, q += lb) { const u32 v = *q; *q = run; run += v; }
        bin_total[(size_t)kw * lb + bin] = run;
    }
}
// pass 1c: LDS-staged scatter, block = group (w, g) as in pass 1a; the group's tiles go through LDS one after the other, the
// per-bin write position carried from tile to tile.  LDS: lh[lb], lbase[lb], gbase[lb], wsum[80], st_ref[ST], st_aux[ST]
template <class D> __global__ __launch_bounds__(1024) void k_s1_scatter(const D* __restrict__ digits, size_t n, u32 gpw, u32 tpb, u32 keyed, u32 lob, u32 lb,
                                                                      u32 ref_base, u32 row_stride, u32 win_lo,
                                                                      const u32* __restrict__ hist1, const u32* __restrict__ part, u32 nblk,
                                                                      const u32* __restrict__ bin_base,
                                                                      u32* __restrict__ p1_ref, unsigned short* __restrict__ p1_lo) {
    extern __shared__ __align__(16) u32 sm[];
    u32 *lh = sm, *lbase = lh + lb, *gbase = lbase + lb, *wsum = gbase + lb, *st_ref = wsum + 80, *st_aux = st_ref + ST;
    const u32 w = blockIdx.x / gpw, g = blockIdx.x % gpw, kw = keyed ? w : 0, tid = threadIdx.x;
    // row of this block inside its key-window's prefix (keyed: the window's own gpw rows; one bucket set: all W * gpw rows)
    const u32 row = keyed ? g : blockIdx.x;
    if (tid < lb) gbase[tid] = bin_base[(size_t)kw * lb + tid] + hist1[(size_t)blockIdx.x * lb + tid]
                               + part[((size_t)kw * nblk + row / S1_RPB) * lb + tid];                          // lb <= 1024
    const u32 lomask = (1u << lob) - 1;
    const size_t glo = (size_t)g * tpb * ST;
    for (u32 tt = 0; tt < tpb; ++tt) {
        const size_t lo = glo + (size_t)tt * ST;
        if (lo >= n) break;
        for (u32 b = tid; b < lb; b += blockDim.x) lh[b] = 0;
        __syncthreads();
        const u32 cnt = (u32)min((size_t)ST, n - lo);
        const D* dg = digits + (size_t)w * n + lo;
        int dv[ST / 1024]; u32 rk[ST / 1024];
#pragma unroll
        for (u32 j = 0; j < ST / 1024; ++j) {
            const u32 e = j * 1024 + tid;
            const int d = e < cnt ? (int)dg[e] : 0;
            dv[j] = d; rk[j] = 0;
            if (d) rk[j] = atomicAdd(&lh[digit_bucket(d) >> lob], 1u);
        }
        __syncthreads();
        u32 tot;
        const u32 mine = tid < lb ? lh[tid] : 0u;
        const u32 ex = block_excl_scan(mine, wsum, &tot);
        if (tid < lb) lbase[tid] = ex;
        __syncthreads();
        const u32 rbase = ref_base + (win_lo + w) * row_stride + (u32)lo;
#pragma unroll
        for (u32 j = 0; j < ST / 1024; ++j) {
            const int d = dv[j];
            if (d) {
                const u32 bucket = digit_bucket(d), bin = bucket >> lob, slot = lbase[bin] + rk[j];
                st_ref[slot] = (rbase + j * 1024 + tid) | (d < 0 ? SIGN_BIT : 0u);
                st_aux[slot] = (bin << 16) | (bucket & lomask);
            }
        }
        __syncthreads();
        for (u32 s = tid; s < tot; s += blockDim.x) {
            const u32 aux = st_aux[s], bin = aux >> 16, ga = gbase[bin] + (s - lbase[bin]);
            p1_ref[ga] = st_ref[s];
            p1_lo[ga] = (unsigned short)aux;
        }
        __syncthreads();
        if (tid < lb) gbase[tid] += mine;                 // the next tile of the group continues where this one ended
    }
}
// pass 2 works on PIECES: piece = up to ST consecutive entries of one bin's segment, every bin has at least one (so
// its offsets get written).  A skewed scalar distribution (a witness with 25 % ones puts a quarter of all entries in
// one bucket) therefore spreads over many blocks; positions inside a bucket are claimed with one global atomic per
// (piece, non-empty sub-bucket) -- the order of a bucket's entries is not deterministic, their sum is.
// piece_base[bin] = first piece of bin, piece_base[hbins] = number of pieces
static __global__ __launch_bounds__(1024) void k_s2_pieces(const u32* __restrict__ bin_base, u32 hbins, u32* __restrict__ piece_base) {
    __shared__ u32 part[1024];
    const u32 t = threadIdx.x, T = blockDim.x;
    const u32 per = (hbins + T - 1) / T;
    const u32 lo = min(t * per, hbins), hi = min(lo + per, hbins);
    u32 s = 0;
    for (u32 k = lo; k < hi; ++k) { const u32 len = bin_base[k + 1] - bin_base[k]; s += len ? (len + ST - 1) / ST : 1u; }
    part[t] = s;
    __syncthreads();
    for (u32 d = 1; d < T; d <<= 1) {
        u32 v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u32 run = part[t] - s;
    for (u32 k = lo; k < hi; ++k) { const u32 len = bin_base[k + 1] - bin_base[k]; piece_base[k] = run; run += len ? (len + ST - 1) / ST : 1u; }
    if (t == T - 1) piece_base[hbins] = part[t];
}
// block -> (bin, first entry, entry count) of its piece; false if the block is beyond the last piece
__device__ __forceinline__ bool s2_piece(const u32* __restrict__ bin_base, const u32* __restrict__ piece_base, u32 hbins,
                                         u32* bin, u32* e0, u32* cn) {
    const u32 pc = blockIdx.x;
    if (pc >= piece_base[hbins]) return false;
    u32 lo = 0, hi = hbins;                   // largest bin with piece_base[bin] <= pc
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (piece_base[mid] <= pc) lo = mid; else hi = mid; }
    const u32 s0 = bin_base[lo] + (pc - piece_base[lo]) * ST, s1 = bin_base[lo + 1];
    *bin = lo; *e0 = s0; *cn = s1 > s0 ? min(ST, s1 - s0) : 0u;
    return true;
}
// pass 2a, block = piece: sub-bucket histogram in LDS, merged into counts[] (zeroed by the host) with global atomics
static __global__ __launch_bounds__(1024) void k_s2_count(const u32* __restrict__ bin_base, const u32* __restrict__ piece_base, u32 hbins,
                                                          const unsigned short* __restrict__ p1_lo, u32 lob, u32* __restrict__ counts) {
    __shared__ u32 h[1024];
    u32 bin, e0, cn;
    if (!s2_piece(bin_base, piece_base, hbins, &bin, &e0, &cn)) return;
    const u32 tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    for (u32 e = tid; e < cn; e += blockDim.x) atomicAdd(&h[p1_lo[e0 + e]], 1u);
    __syncthreads();
    if (h[tid]) atomicAdd(&counts[((size_t)bin << lob) + tid], h[tid]);
}
// pass 2b, block = piece.  LDS: base[1024], h2[1024], lb2[1024], gpos[1024], wsum[80], st_ref[ST], st_sub[ST] (u16)
static __global__ __launch_bounds__(1024) void k_s2_scatter(const u32* __restrict__ bin_base, const u32* __restrict__ piece_base, u32 hbins,
                                                            const u32* __restrict__ p1_ref, const unsigned short* __restrict__ p1_lo, u32 lob,
                                                            const u32* __restrict__ counts, u32* __restrict__ cursor,
                                                            u32* __restrict__ sorted, u32* __restrict__ offsets) {
    extern __shared__ __align__(16) u32 sm[];
    u32 *base = sm, *h2 = base + 1024, *lb2 = h2 + 1024, *gpos = lb2 + 1024, *wsum = gpos + 1024, *st_ref = wsum + 80;
    unsigned short* st_sub = (unsigned short*)(st_ref + ST);
    u32 bin, e0, cn;
    if (!s2_piece(bin_base, piece_base, hbins, &bin, &e0, &cn)) return;
    const u32 sb = 1u << lob, tid = threadIdx.x;
    const bool first_piece = blockIdx.x == piece_base[bin];
    const size_t key0 = (size_t)bin << lob;
    u32 tot;
    const u32 ex = block_excl_scan(tid < sb ? counts[key0 + tid] : 0u, wsum, &tot);
    base[tid] = bin_base[bin] + ex;
    if (first_piece && tid < sb) offsets[key0 + tid] = bin_base[bin] + ex;
    h2[tid] = 0;
    __syncthreads();
    u32 rf[ST / 1024], rk[ST / 1024]; unsigned short sl[ST / 1024];
#pragma unroll
    for (u32 j = 0; j < ST / 1024; ++j) {
        const u32 e = j * 1024 + tid;
        rf[j] = 0; rk[j] = 0; sl[j] = 0;
        if (e < cn) { sl[j] = p1_lo[e0 + e]; rf[j] = p1_ref[e0 + e]; rk[j] = atomicAdd(&h2[sl[j]], 1u); }
    }
    __syncthreads();
    const u32 mine = h2[tid];
    const u32 ex2 = block_excl_scan(mine, wsum, &tot);
    lb2[tid] = ex2;
    if (mine) gpos[tid] = base[tid] + atomicAdd(&cursor[key0 + tid], mine);
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < ST / 1024; ++j) {
        if (j * 1024 + tid < cn) { const u32 slot = lb2[sl[j]] + rk[j]; st_ref[slot] = rf[j]; st_sub[slot] = sl[j]; }
    }
    __syncthreads();
    for (u32 s = tid; s < cn; s += blockDim.x) { const u32 sub = st_sub[s]; sorted[gpos[sub] + (s - lb2[sub])] = st_ref[s]; }
}

// ---- HOT KERNEL: segment accumulation -------------------------------------------------------------------
// lane = (window w, segment k): entries [k*seg, min((k+1)*seg, totals[w])) of window w's sorted list.
// minimum resident blocks per CU the hot kernel is compiled for: the paired 14-limb Fp2 instantiations land at 262 registers
// (one wave per SIMD); asking for two waves costs a few spilled values and doubles the occupancy
template <class Cv> struct AccMinBlocks { static constexpr int value = 1; };
template <> struct AccMinBlocks<Curve<Fe2P<Fp377Cfg>>> { static constexpr int value = 2; };
#ifdef DVN_ACC_G1_BLOCKS
template <> struct AccMinBlocks<Curve<Fe<FpCfg>>> { static constexpr int value = DVN_ACC_G1_BLOCKS; };
#endif
template <> struct AccMinBlocks<Curve<Fe2P<Fp381Cfg>>> { static constexpr int value = 2; };
template <class Cv>
__global__ __launch_bounds__(128, AccMinBlocks<Cv>::value) void k_accumulate(const u32* __restrict__ points, size_t pts_off,
                                                    const u32* __restrict__ sorted, const u32* __restrict__ offsets,
                                                    const u32* __restrict__ counts, const u32* __restrict__ totals,
                                                    u32* __restrict__ bucket_sums, u32* __restrict__ edge_first,
                                                    u32* __restrict__ edge_last, size_t n, u32 nb, u32 seg, u32 nseg,
                                                    u32 nwin_local) {
    typedef typename Cv::X X;
    typedef typename Cv::A A;
    const size_t g = lgtid<Cv>();
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
        if (b_start >= pos0 && b_end <= pos1) dst = bucket_sums + ((size_t)w * nb + b) * AccImg<Cv>::W;
        else if (b_start <= pos0) dst = edge_first + ((size_t)w * nseg + k) * AccImg<Cv>::W;
        else dst = edge_last + ((size_t)w * nseg + k) * AccImg<Cv>::W;
        st_acc<Cv>(acc, dst);
        (void)end_x;
    };
#ifndef DVN_ACC_PREFETCH
#define DVN_ACC_PREFETCH 1      // measured (bench, 2^22): 0 -> 780, 1 -> 787-791, 2 (+ 3-wave bound: 168 VGPRs, 32 B scratch) -> 775-790 Mscalar/s
#endif
#if DVN_ACC_PREFETCH == 0
    for (u32 x = pos0; x < pos1; ++x) {
        if (x == b_end) {                       // bucket b is exhausted: store its run, move to the next non-empty bucket
            flush(x);
            acc = Cv::infinity();
            do { ++b; } while (cnt[b] == 0);
            b_start = off[b]; b_end = b_start + cnt[b];
        }
        const u32 e = src[x];
        A p = ld_affine<Cv>(points + (pts_off + (size_t)(e & ~SIGN_BIT)) * Cv::AW);
        acc = Cv::acc_add_mixed_signed(acc, p, (e & SIGN_BIT) != 0);
    }
#elif DVN_ACC_PREFETCH == 1
    // the sorted entry of iteration x + 1 is fetched before the addition of iteration x: the point gather of the next iteration
    // no longer waits for its index
    u32 e = src[pos0];
    for (u32 x = pos0; x < pos1; ++x) {
        if (x == b_end) {
            flush(x);
            acc = Cv::infinity();
            do { ++b; } while (cnt[b] == 0);
            b_start = off[b]; b_end = b_start + cnt[b];
        }
        const u32 e_next = src[x + 1 < pos1 ? x + 1 : x];
        A p = ld_affine<Cv>(points + (pts_off + (size_t)(e & ~SIGN_BIT)) * Cv::AW);
        acc = Cv::acc_add_mixed_signed(acc, p, (e & SIGN_BIT) != 0);
        e = e_next;
    }
#else
    // ... and the next point itself (raw words) is in flight during the addition
    u32 e = src[pos0];
    u32 raw[Cv::AW];
    load_words16(points + (pts_off + (size_t)(e & ~SIGN_BIT)) * Cv::AW, raw, Cv::AW);
    for (u32 x = pos0; x < pos1; ++x) {
        if (x == b_end) {
            flush(x);
            acc = Cv::infinity();
            do { ++b; } while (cnt[b] == 0);
            b_start = off[b]; b_end = b_start + cnt[b];
        }
        A p = Cv::load_affine_internal(raw);
        if (e & SIGN_BIT) p = Cv::neg(p);
        const u32 e_next = src[x + 1 < pos1 ? x + 1 : x];
        load_words16(points + (pts_off + (size_t)(e_next & ~SIGN_BIT)) * Cv::AW, raw, Cv::AW);
        acc = Cv::acc_add_mixed(acc, p);
        e = e_next;
    }
#endif
    flush(pos1);
}

// lane per bucket: empty -> infinity; spanning <= FIX_LIGHT segments -> stitch serially; more -> heavy list
constexpr u32 FIX_LIGHT = 24;
constexpr u32 FIX_GIANT = 1536, GIANT_PIECES = 64;      // see k_giant_partial
template <class Cv>
__global__ __launch_bounds__(128) void k_fix(const u32* __restrict__ offsets, const u32* __restrict__ counts,
                                             u32* __restrict__ bucket_sums, const u32* __restrict__ edge_first,
                                             const u32* __restrict__ edge_last, u32* __restrict__ heavy_list,
                                             u32* __restrict__ heavy_count, u32 nwin_local, u32 nb, u32 seg, u32 nseg,
                                             u32 heavy_cap) {
    typedef typename Cv::X X;
    const size_t g = lgtid<Cv>();
    if (g >= (size_t)nwin_local * nb) return;
    const u32 w = (u32)(g / nb);
    const u32 c = counts[g];
    u32* dst = bucket_sums + g * AccImg<Cv>::W;
    if (c == 0) { st_acc<Cv>(Cv::infinity(), dst); return; }
    const u32 o = offsets[g], e = o + c;
    const u32 ks = o / seg, ke = (e - 1) / seg;
    if (ks == ke) return;                       // complete inside one segment: written by k_accumulate
    if (ke - ks + 1 > FIX_LIGHT) {
        if (threadIdx.x % Lanes<Cv>::LPP == 0) {
            const u32 slot = atomicAdd(heavy_count, 1u);
            if (slot < heavy_cap) heavy_list[slot] = (u32)g;
        }
        return;
    }
    const u32* first = edge_first + (size_t)w * nseg * AccImg<Cv>::W;
    const u32* last = edge_last + (size_t)w * nseg * AccImg<Cv>::W;
    X acc = ld_acc<Cv>(((o % seg) == 0 ? first : last) + (size_t)ks * AccImg<Cv>::W);
    for (u32 k = ks + 1; k <= ke; ++k) acc = Cv::add(acc, ld_acc<Cv>(first + (size_t)k * AccImg<Cv>::W));
    st_acc<Cv>(acc, dst);
}

// block-cooperative sum of points in shared memory (blockDim.x power of two <= 256); result in lane 0.
// Always inlined: as a real function (which the compiler chose for the call-based Fp2 / BW6-761 instantiations) it reaches the
// kernel's dynamic LDS through llvm.amdgcn.dynlds.offset.table indexed by an SGPR the caller does not keep alive across
// calls -- the second call of a loop then addressed LDS out of range (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on gfx950).
template <class Cv> __device__ __forceinline__ typename Cv::X block_sum(typename Cv::X v, u32* sh /* blockDim.x * XW words */) {
    const u32 t = ltid<Cv>();
    for (u32 stride = lbdim<Cv>() >> 1; stride >= 1; stride >>= 1) {
        if (t >= stride && t < 2 * stride) st_xyzz<Cv>(v, sh + (size_t)(t - stride) * Cv::XW);
        __syncthreads();
        if (t < stride) v = Cv::add(v, ld_xyzz<Cv>(sh + (size_t)t * Cv::XW));
        __syncthreads();
    }
    return v;
}

template <class Cv>
__global__ __launch_bounds__(256 * Lanes<Cv>::LPP) void k_fix_heavy(const u32* __restrict__ offsets, const u32* __restrict__ counts,
                                                   u32* __restrict__ bucket_sums, const u32* __restrict__ edge_first,
                                                   const u32* __restrict__ edge_last, const u32* __restrict__ heavy_list,
                                                   const u32* __restrict__ heavy_count, u32 nb, u32 seg, u32 nseg, u32 heavy_cap,
                                                   u32* __restrict__ giant_list, u32* __restrict__ giant_count, u32 giant_cap, u32 giant_thresh) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 nheavy = min(*heavy_count, heavy_cap);
    for (u32 h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const u32 g = heavy_list[h];
        const u32 w = g / nb;
        const u32 o = offsets[g], e = o + counts[g];
        const u32 ks = o / seg, ke = (e - 1) / seg;
        if (ke - ks + 1 > giant_thresh) {         // a bucket of this size gets GIANT_PIECES blocks (k_giant_*)
            if (threadIdx.x == 0) { const u32 slot = atomicAdd(giant_count, 1u); if (slot < giant_cap) giant_list[slot] = g; }
            continue;
        }
        const u32* first = edge_first + (size_t)w * nseg * AccImg<Cv>::W;
        const u32* last = edge_last + (size_t)w * nseg * AccImg<Cv>::W;
        X acc = Cv::infinity();
        for (u32 k = ks + ltid<Cv>(); k <= ke; k += lbdim<Cv>()) {
            const u32* src = (k == ks && (o % seg) != 0) ? last : first;
            acc = Cv::add(acc, ld_acc<Cv>(src + (size_t)k * AccImg<Cv>::W));
        }
        acc = block_sum<Cv>(acc, sh);
        if (ltid<Cv>() == 0) st_acc<Cv>(acc, bucket_sums + (size_t)g * AccImg<Cv>::W);
        __syncthreads();
    }
}

// Giant buckets (more than FIX_GIANT segments: a witness with 25 % ones, or the 2-bit top window of a 254-bit scalar cut
// into 18- or 21-bit windows, puts a quarter of ALL entries into one bucket): GIANT_PIECES blocks per bucket sum slices of
// its segment partials, a second small kernel adds the slices up.
template <class Cv>
__global__ __launch_bounds__(256 * Lanes<Cv>::LPP) void k_giant_partial(const u32* __restrict__ offsets, const u32* __restrict__ counts,
                                                       const u32* __restrict__ edge_first, const u32* __restrict__ edge_last,
                                                       const u32* __restrict__ giant_list, const u32* __restrict__ giant_count, u32 giant_cap,
                                                       u32 nb, u32 seg, u32 nseg, u32* __restrict__ giant_part) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 items = min(*giant_count, giant_cap) * GIANT_PIECES;
    for (u32 it = blockIdx.x; it < items; it += gridDim.x) {
        const u32 g = giant_list[it / GIANT_PIECES], piece = it % GIANT_PIECES;
        const u32 w = g / nb;
        const u32 o = offsets[g], e = o + counts[g];
        const u32 ks = o / seg, ke = (e - 1) / seg;
        const u32 per = (ke - ks + GIANT_PIECES) / GIANT_PIECES;          // ceil((ke - ks + 1) / GIANT_PIECES)
        const u32 k0 = ks + piece * per, k1 = min(k0 + per, ke + 1);
        const u32* first = edge_first + (size_t)w * nseg * AccImg<Cv>::W;
        const u32* last = edge_last + (size_t)w * nseg * AccImg<Cv>::W;
        X acc = Cv::infinity();
        for (u32 k = k0 + ltid<Cv>(); k < k1; k += lbdim<Cv>()) {
            const u32* src = (k == ks && (o % seg) != 0) ? last : first;
            acc = Cv::add(acc, ld_acc<Cv>(src + (size_t)k * AccImg<Cv>::W));
        }
        acc = block_sum<Cv>(acc, sh);
        if (ltid<Cv>() == 0) st_xyzz<Cv>(acc, giant_part + (size_t)it * Cv::XW);
        __syncthreads();
    }
}
template <class Cv>
__global__ __launch_bounds__(64 * Lanes<Cv>::LPP) void k_giant_final(const u32* __restrict__ giant_list, const u32* __restrict__ giant_count, u32 giant_cap,
                                                    const u32* __restrict__ giant_part, u32* __restrict__ bucket_sums) {
    extern __shared__ __align__(16) u32 sh[];
    const u32 ng = min(*giant_count, giant_cap);
    for (u32 h = blockIdx.x; h < ng; h += gridDim.x) {             // logical block size == GIANT_PIECES
        typename Cv::X acc = ld_xyzz<Cv>(giant_part + ((size_t)h * GIANT_PIECES + ltid<Cv>()) * Cv::XW);
        acc = block_sum<Cv>(acc, sh);
        if (ltid<Cv>() == 0) st_acc<Cv>(acc, bucket_sums + (size_t)giant_list[h] * AccImg<Cv>::W);
        __syncthreads();
    }
}

// ---- bucket reduction: window sum = sum_{b=1..nb} b * B_b -------------------------------------------------
// block handles blockDim.x * L consecutive buckets of one window; thread handles L of them.
template <class Cv>
__global__ __launch_bounds__(256 * Lanes<Cv>::LPP) void k_bucket_reduce(const u32* __restrict__ bucket_sums, u32* __restrict__ partials,
                                                       u32 nb, u32 L, u32 blocks_per_window) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 w = blockIdx.x / blocks_per_window, blk = blockIdx.x % blocks_per_window;
    const u32 first = (blk * lbdim<Cv>() + ltid<Cv>()) * L;     // 0-based index of this thread's first bucket
    X run = Cv::infinity(), tot = Cv::infinity();
    if (first < nb) {
        const u32 last = min(first + L, nb);
        const u32* base = bucket_sums + (size_t)w * nb * AccImg<Cv>::W;
        for (u32 b = last; b-- > first;) {
            run = Cv::add(run, ld_acc<Cv>(base + (size_t)b * AccImg<Cv>::W));
            tot = Cv::add(tot, run);
        }
        // tot = sum_j (j - first + 1) * B_j ; bucket j (0-based) has weight j + 1 = (j - first + 1) + first
        if (first) tot = Cv::add(tot, Cv::mul_small(run, first));
    }
    tot = block_sum<Cv>(tot, sh);
    if (ltid<Cv>() == 0) st_xyzz<Cv>(tot, partials + (size_t)blockIdx.x * Cv::XW);
}
// one small block per window: tree-sum of the window's partial results
template <class Cv>
__global__ __launch_bounds__(64 * Lanes<Cv>::LPP) void k_window_finalize(const u32* __restrict__ partials, u32* __restrict__ window_sums, u32 nwin_local,
                                                        u32 blocks_per_window) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 w = blockIdx.x;
    X acc = Cv::infinity();
    for (u32 i = ltid<Cv>(); i < blocks_per_window; i += lbdim<Cv>())
        acc = Cv::add(acc, ld_xyzz<Cv>(partials + ((size_t)w * blocks_per_window + i) * Cv::XW));
    acc = block_sum<Cv>(acc, sh);
    if (ltid<Cv>() == 0) st_xyzz<Cv>(acc, window_sums + (size_t)w * Cv::XW);
    (void)nwin_local;
}

// ---- bucket reduction of ONE large bucket set (expanded point sets: 2^14 .. 2^20 buckets), without running sums --------
// sum_b (b + 1) B_b with b = hi * RL + lo (RL = 2^kl columns, RH = nb / RL rows) splits into two plain marginal sums,
//     RL * sum_hi hi * R_hi  +  sum_lo (lo + 1) * C_lo,      R_hi = sum_lo B[hi][lo],   C_lo = sum_hi B[hi][lo],
// and a weighted sum over <= 1024 marginals splits by the bits of the weight into plain subset sums,
//     sum_e w_e M_e = sum_j 2^j * (sum over the e whose weight has bit j set of M_e).
// Every step is a plain sum, so the dependent chain is ~25 group operations where chunked running sums with their small scalar
// multiplications need ~60; the 2^j weights are applied on the host (msm_driver.h bitsum_combine), which gets kh + kl + 1
// subset sums in a few pieces each.
// k_marginal_level: ONE level of both marginal sums as a grid-wide step.  Thread g < n_rows adds r_rows neighbours of a row
// (flat groups of consecutive entries), the next n_cols threads add r_cols entries one row pitch (RL) apart.  Every lane of
// every wave adds; the levels are separate launches (groups of four: ceil(log4) launches, three dependent additions each)
// where an in-block tree leaves half of the lanes idle at every step and pays two barriers per step.
// Level 1 reads the raw accumulator images of the bucket array (src_acc = 1), later levels XYZZ images.
constexpr u32 MARG_THREADS = 128, MARG_RADIX = 4;
template <class Cv>
__global__ __launch_bounds__(MARG_THREADS * Lanes<Cv>::LPP) void k_marginal_level(const u32* __restrict__ src_rows, const u32* __restrict__ src_cols, int src_acc,
                                                                                u32* __restrict__ dst_rows, u32* __restrict__ dst_cols,
                                                                                u32 n_rows, u32 r_rows, u32 n_cols, u32 r_cols, u32 kl) {
    typedef typename Cv::X X;
    const size_t g = lgtid<Cv>();
    if (g >= (size_t)n_rows + n_cols) return;
    const u32* src; u32* dst; size_t first, stride; u32 cnt;
    if (g < n_rows) { src = src_rows; dst = dst_rows + g * Cv::XW; first = g * r_rows; stride = 1; cnt = r_rows; }
    else {
        const size_t i = g - n_rows, h = i >> kl, lo = i & ((1u << kl) - 1);
        src = src_cols; dst = dst_cols + i * Cv::XW; first = ((h * r_cols) << kl) + lo; stride = (size_t)1 << kl; cnt = r_cols;
    }
    auto load = [&](size_t e) { return src_acc ? ld_acc<Cv>(src + e * AccImg<Cv>::W) : ld_xyzz<Cv>(src + e * Cv::XW); };
    X acc = load(first);
    for (u32 k = 1; k < cnt; ++k) acc = Cv::add(acc, load(first + k * stride));
    st_xyzz<Cv>(acc, dst);
}
// k_marginal_bitsums: the subset sums of the marginals, BITSUM_SPAN marginals per block.  Blocks [0, kh * sr) take bit
// j = block / sr of the row index hi (weight of R_hi), the next (kl + 1) * sc blocks bit j of lo + 1 (weight of C_lo);
// sr = ceil(RH / BITSUM_SPAN), sc = ceil(RL / BITSUM_SPAN).  out[block] receives the piece.
constexpr u32 BITSUM_THREADS = 256, BITSUM_SPAN = dvn_host::BITSUM_SPAN;      // marginals per piece: the host's combine uses the same span
template <class Cv>
__global__ __launch_bounds__(BITSUM_THREADS * Lanes<Cv>::LPP) void k_marginal_bitsums(const u32* __restrict__ marg, u32* __restrict__ out, u32 nb, u32 kl) {
    typedef typename Cv::X X;
    extern __shared__ __align__(16) u32 sh[];
    const u32 RL = 1u << kl, RH = nb >> kl;
    u32 kh = 0; while ((1u << kh) < RH) ++kh;
    const u32 sr = (RH + BITSUM_SPAN - 1) / BITSUM_SPAN, sc = (RL + BITSUM_SPAN - 1) / BITSUM_SPAN;
    const u32 t = ltid<Cv>(), T = lbdim<Cv>();
    u32 j, s, cnt, bias; const u32* src;
    if (blockIdx.x < kh * sr) { j = blockIdx.x / sr; s = blockIdx.x % sr; cnt = RH; bias = 0; src = marg; }
    else { const u32 b = blockIdx.x - kh * sr; j = b / sc; s = b % sc; cnt = RL; bias = 1; src = marg + (size_t)RH * Cv::XW; }
    X acc = Cv::infinity();
    const u32 hi = min(cnt, (s + 1) * BITSUM_SPAN);
    for (u32 e = s * BITSUM_SPAN + t; e < hi; e += T)
        if (((e + bias) >> j) & 1) acc = Cv::add(acc, ld_xyzz<Cv>(src + (size_t)e * Cv::XW));
    acc = block_sum<Cv>(acc, sh);
    if (t == 0) st_xyzz<Cv>(acc, out + (size_t)blockIdx.x * Cv::XW);
}

// Batched MSM tail: one thread per scalar vector combines its nwin window sums MSB first (gnark's
// `for j = last..0: acc = 2^c acc + W_j`) on the device -- with B results the sequential chain is no longer worth a
// round trip to the host per result.  out: B internal xyzz images.
template <class Cv>
__global__ void k_batch_horner(const u32* __restrict__ window_sums, u32 nwin, u32 c, u32 batch, u32* __restrict__ out) {
    typedef typename Cv::X X;
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    X acc = Cv::infinity();
    for (int w = (int)nwin - 1; w >= 0; --w) {
        for (u32 k = 0; k < c; ++k) acc = Cv::dbl(acc);
        acc = Cv::add(acc, ld_xyzz<Cv>(window_sums + ((size_t)b * nwin + w) * Cv::XW));
    }
    st_xyzz<Cv>(acc, out + (size_t)b * Cv::XW);
}

}  // namespace dvn
