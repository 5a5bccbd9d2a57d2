// Radix-2 NTT over BN254 Fr on gfx950 with gnark-crypto's fft.Domain semantics:
//   DIF: natural in -> bit-reversed out, butterfly (a, b) -> (a + b, (a - b) * w)
//   DIT: bit-reversed in -> natural out, butterfly (a, b) -> (a + b * w, a - b * w)
//   FFT(OnCoset) pre-scales by g^i (bit-reversed index when DIT); FFTInverse post-scales by 1/n (and g^-i,
//   bit-reversed index when DIF).  No implicit bit reversal anywhere.
// Used by groth16 computeH (7 transforms per proof; /root/reference/prover/prover_cpu.go:37 -> gnark prove.go).
//
// One launch = one PASS of up to 8 consecutive radix-2 stages on LDS-resident tiles: a tile is I (<= 8)
// neighbouring sub-problems of 2^K elements, so every global access is a run of I * 32 B = 256 B and the
// 2^K * I-element tile (2048 elements of nine limbs = 72 KiB of LDS) sees K stages for one HBM read + one HBM write.  Twiddles come from
// per-stage tables table_s[k] = w^(k * 2^s) in HBM (contiguous in k, i.e. coalesced across a tile row); the
// late, small tables stay in L2 / Infinity Cache.  A table entry is stored as the NL 29-bit limbs the multiplier takes (limb
// planes over the whole table, the layout of the LDS tile), not as packed words: no bit slicing per twiddle.
//
// Coset transforms need no scaling pass where the butterfly network can absorb it: a DIT network whose stage-s twiddles are
// g^(2^s) * w^(k 2^s) evaluates on the coset g<w> directly (P(g x) = Pe(g^2 x^2) + g x Po(g^2 x^2)), and a DIF network with
// g^-(2^s) * w^-(k 2^s) returns the inverse transform already multiplied by g^-j.  The domain holds those two extra table sets.
//
// Data words are gnark Montgomery words (x * 2^256), read as the internal form of x * 2^-5; the transform is
// linear so no conversion pass is needed (bn254_field.h header).  Twiddles / scale factors are true internal
// forms.
#pragma once
#include "bn254_field.h"

namespace dvn {

// one element = S::EW words (8 for the 253..255-bit scalar fields, 12 for BW6-761's 377-bit one)
template <class S> __device__ __forceinline__ Fe<S> ld_fr(const u32* __restrict__ p) {
    u32 w[S::EW];
#pragma unroll
    for (int i = 0; i < S::EW; i += 4) { const uint4 v = *reinterpret_cast<const uint4*>(p + i); w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w; }
    return fe_unpack<S>(w);
}
// weak form -> canonical -> EW words
template <class S> __device__ __forceinline__ void st_fr(const Fe<S>& v, u32* __restrict__ p) {
    u32 w[S::EW];
    fe_pack<S>(fe_canonical<S>(v), w);
#pragma unroll
    for (int i = 0; i < S::EW; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(w[i], w[i + 1], w[i + 2], w[i + 3]);
}
// packs the value as it is (no canonicalisation): limbs normalised, value below 2^(32 EW)
template <class S> __device__ __forceinline__ void st_fr_weak(const Fe<S>& v, u32* p) {
    u32 w[S::EW];
    fe_pack<S>(v, w);
#pragma unroll
    for (int i = 0; i < S::EW; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(w[i], w[i + 1], w[i + 2], w[i + 3]);
}

// LDS slots hold an element as its NL 29-bit limbs, NOT as packed words: a round reads and writes the limbs the field
// routines work on, so the per-round unpack (word -> limb bit slicing) and pack disappear from the inner loop.  Layout: limb
// quads in planes of 16 bytes per element (plane k at k * E * 16), the NL % 4 remaining limbs in planes of 4 bytes per element
// -- consecutive lanes touch consecutive 16-byte (4-byte) cells of a plane, so the accesses are free of bank conflicts
// where 32-byte packed slots read as two 16-byte halves were not.  NL = 9: 36 bytes per element, a 2048-element tile is 72 KiB.
template <class S> __device__ __forceinline__ Fe<S> lds_ld(const u32* lds, u32 e, u32 E) {
    constexpr int NQ = S::NL / 4, NR = S::NL % 4;
    Fe<S> v;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const uint4 q = *reinterpret_cast<const uint4*>(lds + ((size_t)k * E + e) * 4);
        v.v[4 * k] = q.x; v.v[4 * k + 1] = q.y; v.v[4 * k + 2] = q.z; v.v[4 * k + 3] = q.w;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) v.v[4 * NQ + r] = lds[(size_t)NQ * E * 4 + (size_t)r * E + e];
    return v;
}
template <class S> __device__ __forceinline__ void lds_st(u32* lds, u32 e, u32 E, const Fe<S>& v) {
    constexpr int NQ = S::NL / 4, NR = S::NL % 4;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        *reinterpret_cast<uint4*>(lds + ((size_t)k * E + e) * 4) = make_uint4(v.v[4 * k], v.v[4 * k + 1], v.v[4 * k + 2], v.v[4 * k + 3]);
#pragma unroll
    for (int r = 0; r < NR; ++r) lds[(size_t)NQ * E * 4 + (size_t)r * E + e] = v.v[4 * NQ + r];
}

// out[k] = base^(k) * first, k < count; thread handles `chunk` consecutive entries.  base/first internal form.
template <class S> __global__ void k_powers(u32* __restrict__ out, const u32* __restrict__ base_first, size_t count, u32 chunk) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * chunk;
    if (lo >= count) return;
    size_t hi = lo + chunk < count ? lo + chunk : count;
    Fe<S> base = ld_fr<S>(base_first), cur = fe_mul<S>(fe_pow_u64<S>(base, (u64)lo), ld_fr<S>(base_first + S::EW));
    for (size_t k = lo; k < hi; ++k) { st_fr<S>(cur, out + k * S::EW); cur = fe_mul<S>(cur, base); }
}

// the same progression written as limb planes: entry e0 + k of a table of E entries (lds_st layout), canonical limbs
template <class S> __global__ void k_powers_planes(u32* __restrict__ out, u32 E, u32 e0, const u32* __restrict__ base_first, size_t count, u32 chunk) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * chunk;
    if (lo >= count) return;
    size_t hi = lo + chunk < count ? lo + chunk : count;
    Fe<S> base = ld_fr<S>(base_first), cur = fe_mul<S>(fe_pow_u64<S>(base, (u64)lo), ld_fr<S>(base_first + S::EW));
    for (size_t k = lo; k < hi; ++k) { lds_st<S>(out, e0 + (u32)k, E, fe_canonical<S>(cur)); cur = fe_mul<S>(cur, base); }
}

struct NttScale {
    const u32* lo;     // lo[j]  = f^j * c,            j < 2^lo_bits   (mode 2); lo[0] = constant (mode 1)
    const u32* hi;     // hi[j]  = f^(j << lo_bits),   j < n >> lo_bits
    u32 lo_bits;
    int mode;          // 0 none, 1 constant, 2 power table, 3 (at load only) computeH's pointwise step: v = v * lo[i] * aux[0] - hi[i] * aux[1]
    int bitrev;        // exponent = bitrev(i) instead of i
    int when;          // 1 = at load (pre-scale), 2 = at store (post-scale)
    const u32* aux;    // mode 3: the two constants (2 elements)
};

template <class S> __device__ __forceinline__ Fe<S> ntt_scale_factor(const NttScale& sc, u32 i, u32 log_n) {
    if (sc.mode == 1) return ld_fr<S>(sc.lo);
    const u32 e = sc.bitrev ? (__brev(i) >> (32 - log_n)) : i;
    Fe<S> f = ld_fr<S>(sc.lo + (size_t)(e & ((1u << sc.lo_bits) - 1)) * S::EW);
    if (log_n > sc.lo_bits) f = fe_mul<S>(f, ld_fr<S>(sc.hi + (size_t)(e >> sc.lo_bits) * S::EW));
    return f;
}

// In-pass invariants: limbs in LDS are normalised (a slot is the NL limbs).  DIF rounds keep values below 4p (ntt_red brings
// a value below 8p back under it): their sums would otherwise quadruple per round.  DIT rounds do not reduce at all: a
// round's outputs are x +- t +- u (+ 2p per subtraction) with t, u fresh products below 2p, so a value grows by at most 4p
// per round -- at most 2p + 4 * 4p = 18p over the 4 rounds of a pass, inside the 2^(BITS + 6) that fe_reduce_full takes at
// the store and far inside what a multiplication by a canonical twiddle accepts.
template <class S> __device__ __forceinline__ Fe<S> ntt_red(const Fe<S>& x) { return fe_csub<S, 4>(x); }
template <class S> struct NttBetween { static constexpr bool RAW_4P = S::BITS + 2 <= 32 * S::EW; };
// wide fields (NL > 9) leave less headroom in fe_mul's 64-bit columns: their raw operands are built from normalised sums
template <class S> struct NttRaw {
    static constexpr bool TIGHT = S::NL > 9;
    static constexpr int SH_SUMS = TIGHT ? 29 : 30;       // limb bound of the sums S0, S1 below
    static_assert(fe_mul_raw_fits<S>((1ull << SH_SUMS) + (1ull << SH_SUMS) + (1ull << LB)), "p01 operand");
    static_assert(fe_mul_raw_fits<S>(3ull << LB), "operands built from two normalised values");
};

// stages [sa, sb) of a 2^log_n transform; dit = 0: ascending s (DIF), dit = 1: descending s (DIT); last = 1 on the pass
// that finishes the transform (its store is canonical).
// Two stages per LDS round trip (radix-4 units held in registers); values in LDS are kept below 4p with normalised limbs.
// Differences feed the twiddle multiplication with NO carry pass (fe_sub_raw: the twiddle is canonical, so the product's
// 64-bit columns have room for limbs up to 2.5 * 2^30 and the result is below 2p for operands up to 24p); only what is
// stored without passing through a multiplication pays a carry pass and conditional subtractions.
// triv = 1: the tables are the plain powers of w and this pass holds the last two stages (sb == log_n): their radix-4 round
// multiplies by 1, 1, 1 and w^(n/4) only, so it runs with one multiplication instead of four.
template <class S> __global__ __launch_bounds__(512, (S::NL <= 9 ? 4 : 2)) void k_ntt_pass(u32* __restrict__ data, const u32* __restrict__ tw, u32 log_n,
                                                  u32 sa, u32 sb, int dit, int last, u32 log_i, int triv, NttScale sc) {
    typedef S C;
    typedef Fe<S> Fr;
    extern __shared__ __align__(16) u32 lds[];
    const u32 K = sb - sa, Lb = log_n - sb, I = 1u << log_i;
    const u32 E = I << K;
    const u32 n = 1u << log_n;
    const u32 sp0 = blockIdx.x << log_i;                 // first sub-problem of this tile
    const u32 inner_mask = (1u << Lb) - 1;
    // ---- load
    for (u32 le = threadIdx.x; le < E; le += blockDim.x) {
        const u32 q = le & (I - 1), D = le >> log_i, sp = sp0 + q;
        const u32 gi = ((sp >> Lb) << (Lb + K)) | (D << Lb) | (sp & inner_mask);
        Fr v = ld_fr<S>(data + (size_t)gi * S::EW);
        if (sc.mode == 3) {
            const Fr ab = fe_mul<C>(fe_mul<C>(v, ld_fr<S>(sc.lo + (size_t)gi * S::EW)), ld_fr<S>(sc.aux));
            v = fe_sub<C>(ab, fe_mul<C>(ld_fr<S>(sc.hi + (size_t)gi * S::EW), ld_fr<S>(sc.aux + S::EW)));
        } else if (sc.mode && sc.when == 1) v = fe_mul<C>(v, ntt_scale_factor<S>(sc, gi, log_n));
        else if (triv && (dit || sa == 0)) v = fe_csub<C, 4>(v);      // any EW-word input is below 8p: the multiplication-free round wants < 4p
        lds_st<S>(lds, le, E, v);
    }
    __syncthreads();
    // ---- radix-4 rounds: executed stages t and t + 1
    u32 t = 0;
    for (; t + 1 < K; t += 2) {
        // bit positions inside D of the two stages; dlo is the lower one
        const u32 dlo = dit ? t : (K - 2 - t);
        const u32 s_first = dit ? (sb - 1 - t) : (sa + t);           // first executed stage
        const u32 s_second = dit ? (s_first - 1) : (s_first + 1);
        const u32 tab1 = n - (n >> s_first), tab2 = n - (n >> s_second);     // first entries of the two stages' tables
        if (triv && dlo == 0 && Lb == 0 && (dit ? t == 0 : t + 2 == K)) {
            // the last two stages of the transform (executed first by DIT): twiddles 1, 1, 1 and w4 = w^(n/4)
            const Fr w4 = lds_ld<S>(tw, (dit ? tab2 : tab1) + 1, n);      // second entry of stage log_n - 2
            for (u32 u = threadIdx.x; u < (E >> 2); u += blockDim.x) {
                const u32 q = u & (I - 1), Db = (u >> log_i) << 2;
                const u32 p00 = (Db << log_i) | q, p01 = ((Db | 1u) << log_i) | q, p10 = ((Db | 2u) << log_i) | q, p11 = ((Db | 3u) << log_i) | q;
                const Fr x00 = lds_ld<S>(lds, p00, E), x01 = lds_ld<S>(lds, p01, E), x10 = lds_ld<S>(lds, p10, E), x11 = lds_ld<S>(lds, p11, E);
                if (!dit) {          // inputs < 4p; outputs < 16p, taken by the store's full reduction (or its scale multiplication)
                    const Fr S0 = fe_add_lazy<C>(x00, x10), S1 = fe_add_lazy<C>(x01, x11);
                    const Fr D0 = fe_sub_lazy<C, 4>(x00, x10), B1 = fe_mul<C>(fe_sub_raw<C, 8, LB>(x01, x11), w4);
                    lds_st<S>(lds, p00, E, fe_add_lazy<C>(S0, S1));
                    lds_st<S>(lds, p01, E, fe_sub_lazy<C, 8>(S0, S1));
                    lds_st<S>(lds, p10, E, fe_add_lazy<C>(D0, B1));
                    lds_st<S>(lds, p11, E, fe_sub_lazy<C, 2>(D0, B1));
                } else {             // inputs < 4p; outputs < 16p, and each later round adds at most 4p
                    const Fr a00 = fe_add_lazy<C>(x00, x01), a01 = fe_sub_lazy<C, 4>(x00, x01);
                    const Fr a10 = fe_add_lazy<C>(x10, x11), u1 = fe_mul<C>(fe_sub_raw<C, 8, LB>(x10, x11), w4);
                    lds_st<S>(lds, p00, E, fe_add_lazy<C>(a00, a10));
                    lds_st<S>(lds, p01, E, fe_add_lazy<C>(a01, u1));
                    lds_st<S>(lds, p10, E, fe_sub_lazy<C, 8>(a00, a10));
                    lds_st<S>(lds, p11, E, fe_sub_lazy<C, 2>(a01, u1));
                }
            }
            __syncthreads();
            continue;
        }
        for (u32 u = threadIdx.x; u < (E >> 2); u += blockDim.x) {
            const u32 q = u & (I - 1), dq = u >> log_i;
            const u32 low = dq & ((1u << dlo) - 1);
            const u32 Db = ((dq >> dlo) << (dlo + 2)) | low;
            const u32 inner = (sp0 + q) & inner_mask;
            // x[hi][lo]: hi = bit dlo+1 of D, lo = bit dlo
            const u32 p00 = (Db << log_i) | q, p01 = ((Db | (1u << dlo)) << log_i) | q;
            const u32 p10 = ((Db | (2u << dlo)) << log_i) | q, p11 = ((Db | (3u << dlo)) << log_i) | q;
            const Fr x00 = lds_ld<S>(lds, p00, E), x01 = lds_ld<S>(lds, p01, E), x10 = lds_ld<S>(lds, p10, E), x11 = lds_ld<S>(lds, p11, E);
            if (!dit) {
                // stage s pairs over the HIGH bit (distance 2 << dlo), twiddle depends on the low bit;
                // stage s+1 pairs over the LOW bit, one twiddle for the whole unit
                const Fr wb = lds_ld<S>(tw, tab2 + ((low << Lb) | inner), n);
                // differences first: the four inputs are dead after these four lines (register pressure, not arithmetic, orders this)
                const Fr D0 = fe_sub_raw<C, 8, LB>(x00, x10), D1 = fe_sub_raw<C, 8, LB>(x01, x11);
                Fr S0 = fe_add_nocarry<C>(x00, x10), S1 = fe_add_nocarry<C>(x01, x11);                 // < 8p, limbs < 2^30
                if constexpr (NttRaw<S>::TIGHT) { S0 = fe_normalise<C>(S0); S1 = fe_normalise<C>(S1); }
                lds_st<S>(lds, p00, E, fe_csub<C, 4>(fe_csub<C, 8>(fe_normalise<C>(fe_add_nocarry<C>(S0, S1)))));   // < 16p -> < 4p
                lds_st<S>(lds, p01, E, fe_mul<C>(fe_sub_raw<C, 16, NttRaw<S>::SH_SUMS>(S0, S1), wb));   // operand < 24p
                // the stage-s twiddles are fetched only now: 18 registers that would otherwise be live across the first product
                asm volatile("" ::: "memory");
                const Fr wa0 = lds_ld<S>(tw, tab1 + ((low << Lb) | inner), n);
                const Fr wa1 = lds_ld<S>(tw, tab1 + ((((1u << dlo) | low) << Lb) | inner), n);
                const Fr B0 = fe_mul<C>(D0, wa0), B1 = fe_mul<C>(D1, wa1);
                lds_st<S>(lds, p10, E, fe_add_lazy<C>(B0, B1));                                          // two products: < 4p
                lds_st<S>(lds, p11, E, fe_mul<C>(fe_sub_raw<C, 2, LB>(B0, B1), wb));
            } else {
                // stage s1 pairs over the LOW bit, one twiddle; stage s1-1 pairs over the HIGH bit, twiddle depends on the low bit
                const Fr w1 = lds_ld<S>(tw, tab1 + ((low << Lb) | inner), n);
                const Fr w20 = lds_ld<S>(tw, tab2 + ((low << Lb) | inner), n);
                const Fr w21 = lds_ld<S>(tw, tab2 + ((((1u << dlo) | low) << Lb) | inner), n);
                const Fr t0 = fe_mul<C>(x01, w1), t1 = fe_mul<C>(x11, w1);                      // products: < 2p, normalised
                const Fr a00 = fe_add_nocarry<C>(x00, t0), a01 = fe_sub_raw<C, 2, LB>(x00, t0);  // < 6p, limbs < 2^30 / 1.5 * 2^30
                const Fr a10 = fe_add_nocarry<C>(x10, t1), a11 = fe_sub_raw<C, 2, LB>(x10, t1);
                const Fr u0 = fe_mul<C>(a10, w20), u1 = fe_mul<C>(a11, w21);
                // no reduction: a DIT round adds at most 4p to a value (see the bound above the kernel)
                lds_st<S>(lds, p00, E, fe_normalise<C>(fe_add_nocarry<C>(a00, u0)));
                lds_st<S>(lds, p01, E, fe_normalise<C>(fe_add_nocarry<C>(a01, u1)));
                lds_st<S>(lds, p10, E, fe_normalise<C>(fe_sub_raw<C, 2, LB>(a00, u0)));
                lds_st<S>(lds, p11, E, fe_normalise<C>(fe_sub_raw<C, 2, LB>(a01, u1)));
            }
        }
        __syncthreads();
    }
    // ---- odd stage count: one radix-2 round for the last executed stage
    if (t < K) {
        const u32 s = dit ? (sb - 1 - t) : (sa + t);
        const u32 tl = s - sa;
        const u32 dshift = K - 1 - tl, dist = 1u << dshift;
        const u32 table = n - (n >> s);
        for (u32 u = threadIdx.x; u < (E >> 1); u += blockDim.x) {
            const u32 q = u & (I - 1), dp = u >> log_i;
            const u32 low = dp & (dist - 1);
            const u32 D0 = ((dp >> dshift) << (dshift + 1)) | low, D1 = D0 | dist;
            const u32 j = (low << Lb) | ((sp0 + q) & inner_mask);
            const Fr w = lds_ld<S>(tw, table + j, n);
            const u32 pa = (D0 << log_i) | q, pb = (D1 << log_i) | q;
            const Fr a = lds_ld<S>(lds, pa, E), b = lds_ld<S>(lds, pb, E);
            if (!dit) {
                lds_st<S>(lds, pa, E, ntt_red<S>(fe_add_lazy<C>(a, b)));
                lds_st<S>(lds, pb, E, fe_mul<C>(fe_sub_raw<C, 8, LB>(a, b), w));
            } else {
                const Fr bt = fe_mul<C>(b, w);
                lds_st<S>(lds, pa, E, fe_normalise<C>(fe_add_nocarry<C>(a, bt)));
                lds_st<S>(lds, pb, E, fe_normalise<C>(fe_sub_raw<C, 2, LB>(a, bt)));
            }
        }
        __syncthreads();
    }
    // ---- store
    for (u32 le = threadIdx.x; le < E; le += blockDim.x) {
        const u32 q = le & (I - 1), D = le >> log_i, sp = sp0 + q;
        const u32 gi = ((sp >> Lb) << (Lb + K)) | (D << Lb) | (sp & inner_mask);
        Fr v = lds_ld<S>(lds, le, E);
        if (sc.mode && sc.when == 2) st_fr<S>(fe_mul<C>(v, ntt_scale_factor<S>(sc, gi, log_n)), data + (size_t)gi * S::EW);
        else if (last) st_fr<S>(fe_reduce_full<S>(v), data + (size_t)gi * S::EW);
        else {
            // between the passes of one transform the array need not be canonical: the next pass takes what its first round
            // takes from LDS (DIF: below 4p, when that fits the EW words; otherwise below 2p, DIT's starting bound)
            if (!(NttBetween<S>::RAW_4P && !dit)) v = fe_reduce_full<S>(v);
            st_fr_weak<S>(v, data + (size_t)gi * S::EW);
        }
    }
}

}  // namespace dvn
