// Groth16 host orchestration over the device primitives, generic over the pairing curve:
// the C++ restatement of what the Go shim's hipGroth16Prove does, runnable without a Go toolchain.
//
// Mirrors, step for step, gnark backend/groth16/<curve> prove.go `Prove` (called from
// /root/reference/prover/prover_cpu.go:37 and, through the ICICLE wrapper, prover/prover_gpu.go:33-57) and setup.go
// `Setup` (/root/reference/prover/setup.go:15-28) for a plain sparse R1CS:
//   Setup:  toxic waste -> Lagrange basis at tau -> A_i(tau), B_i(tau), C_i(tau) -> K, Z (bit-reversed) ->
//           fixed-base batch scalar multiplications -> proving key arrays resident in HBM, infinity filtering
//   Prove:  (a, b, c) = (A w, B w, C w)  ->  h = computeH  ->  wire filtering by InfinityA/B  ->
//           Ar, Bs1 (G1 MSM), Bs (G2 MSM), Krs (2 G1 MSMs)  ->  blinding with (r, s)  ->  affine proof
// The R1CS solver proper (hints, emulated arithmetic) stays in gnark on the CPU; here the "solve" is the sparse
// matrix-vector product over an already complete wire assignment.
//
// A suite SU supplies: FrC (scalar config), G1T / G2T, H1 / H2 (host finishing arithmetic), SUITE, r_words(), g1_gen(),
// g2_gen(), domain_create, compute_h and the MSM submit / collect calls (cross-TU for BN254, direct elsewhere).
#pragma once
#include <chrono>
#include <algorithm>
#include <future>
#include <thread>
#include "dvn_internal.h"
#include "msm_kernels.h"
#include "ntt_kernels.h"
#include "ntt_driver.h"
#include "msm_driver.h"

// k = canonical scalar, S::EW words: 64-bit-limb host arithmetic takes 8, the generic one any width
template <class H, class S> static typename H::X host_mul(const typename H::X& p, const u32* k) {
    if constexpr (S::EW == 8) return H::mul_scalar(p, k);
    else return H::mul_scalar_n(p, k, S::EW);
}
template <class SU> struct HostKey { typename SU::H1::X alpha, beta, delta; typename SU::H2::X beta2, delta2; };

// ------------------------------------------------------------------------------------------------------
// types
// ------------------------------------------------------------------------------------------------------
struct Csr {
    std::vector<u32> ptr, idx;      // rows -> (wire, coeff)
    std::vector<u32> coef;          // 8 words per entry, internal Montgomery form (packed)
};
struct dvn_r1cs {
    int suite = 0;                  // scalar field the coefficients and the witness live in (DomainConsts<S>::SUITE)
    u32 n_wires = 0, n_public = 0, n_constraints = 0;
    Csr m[3];                       // A, B, C by constraint row
    std::vector<u32> witness;       // n_wires x 8, gnark Montgomery words (synthetic instances only)
    // one BSB22 commitment (0 = none): wire order [1, inputs, commitment wire, committed private wires, other private wires];
    // only the last `tail` constraints read the commitment wire; public_committed = public wires hashed into the challenge
    u32 commit_wire = 0, n_committed = 0, tail = 0;
    std::vector<u32> public_committed;
};
struct dvn_groth16_pk {
    u32 log_n = 0;
    size_t n = 0, m = 0, n_public = 0, n_constraints = 0;
    dvn_domain* dom = nullptr;
    dvn_points *A = nullptr, *B = nullptr, *K = nullptr, *Z = nullptr, *B2 = nullptr;
    size_t nA = 0, nB = 0;
    u32 *d_idxA = nullptr, *d_idxB = nullptr;           // wires kept after infinity filtering (InfinityA/B)
    int suite = 0;
    void* hk = nullptr;                        // HostKey<SU>: key constants in the host finishing arithmetic
    void (*hk_free)(void*) = nullptr;
    u32 *d_ptr[3] = {nullptr, nullptr, nullptr}, *d_idx[3] = {nullptr, nullptr, nullptr}, *d_coef[3] = {nullptr, nullptr, nullptr};
    u32 *d_w = nullptr, *d_abc = nullptr, *d_wA = nullptr, *d_wB = nullptr;   // per-proof buffers, allocated once
    float ms_h = 0, ms_msm = 0, ms_total = 0;
    // per-proof statistics of the last prove call on this key (dvn_groth16_last_stats_json): device ms of each MSM as its lane
    // measured it (A, B1, K, Z, B2), pairs it consumed, host-to-device bytes of the call, rank / world of a shared proof
    float msm_ms[5] = {0, 0, 0, 0, 0};
    size_t msm_pairs[5] = {0, 0, 0, 0, 0};
    size_t h2d_bytes = 0;
    u32 last_rank = 0, last_world = 1;
    int device = -1;                    // HIP ordinal the key's arrays live on
    u32 scalar_bytes = 0, g1_bytes = 0, g2_bytes = 0;     // image sizes of the key's curve (statistics)
    // BSB22 commitment (n_committed = 0: none): K covers the wires [k_first, m); Basis / BasisSigma the committed ones
    size_t k_first = 0;
    u32 commit_wire = 0, n_committed = 0, tail = 0;
    std::vector<u32> public_committed;
    dvn_points *Basis = nullptr, *BasisSigma = nullptr;
    // imported keys (groth16_pk_import: gnark's own ProvingKey arrays): K's scalars are gathered through d_idxK; no R1CS
    u32 *d_idxK = nullptr, *d_wK = nullptr;
    size_t nK = 0;
    bool imported = false;
    bool cached = false;                // owned by the context's sha256-keyed key cache: dvn_groth16_pk_free leaves it alone
    u32 load_n_public = 0;              // what a key read from an artifact file was loaded with: a cache hit must name the same
    std::vector<u32> load_committed;
    // verifying key as gnark affine Montgomery images (uint64 words): alpha (G1) | beta | gamma | delta (G2) | IC[n_public] (G1:
    // CONSTANT, PUB_i of the Solidity verifier) | Pedersen G | Pedersen GSigmaNeg (G2; only with a commitment)
    std::vector<uint64_t> vk_words;
};

// ------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------

// out[row] = sum_k coef[k] * w[idx[k]]; w are gnark Montgomery words (read as x*2^-5 internal), coef true internal:
// the product is again a gnark Montgomery word.
// Rows [rows, rows_padded) are written as zeros (the vector is zero-padded to the domain size).
template <class S> __global__ void k_r1cs_matvec(const u32* __restrict__ ptr, const u32* __restrict__ idx, const u32* __restrict__ coef,
                                     const u32* __restrict__ w, u32* __restrict__ out, u32 rows, u32 rows_padded) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_padded) return;
    Fe<S> acc = fe_zero<S>();
    if (r >= rows) { st_fr<S>(acc, out + (size_t)r * S::EW); return; }
    for (u32 k = ptr[r]; k < ptr[r + 1]; ++k)
        acc = fe_add<S>(acc, fe_mul<S>(ld_fr<S>(coef + (size_t)k * S::EW), ld_fr<S>(w + (size_t)idx[k] * S::EW)));
    st_fr<S>(acc, out + (size_t)r * S::EW);
}
// The solver's last step with a BSB22 commitment: once the challenge is in its wire, re-evaluate the constraints
// [row_lo, row_hi) in order, w[C_row] = (A_row w)(B_row w) (C rows of the synthetic instance are one wire, coefficient 1).
// One thread: O(tail) work.  c32 = 2^5 in internal form (the words are read as x * 2^-5, see k_r1cs_matvec).
template <class S> __global__ void k_r1cs_patch_rows(const u32* __restrict__ pa, const u32* __restrict__ ia, const u32* __restrict__ ca,
                                                     const u32* __restrict__ pb, const u32* __restrict__ ib, const u32* __restrict__ cb,
                                                     const u32* __restrict__ pc, const u32* __restrict__ ic, u32* __restrict__ w, u32 row_lo, u32 row_hi) {
    if (blockIdx.x || threadIdx.x) return;
    u32 w32[S::EW] = {32u};
    const Fe<S> c32 = fe_from_canonical<S>(fe_unpack<S>(w32));
    for (u32 r = row_lo; r < row_hi; ++r) {
        Fe<S> a = fe_zero<S>(), b = fe_zero<S>();
        for (u32 k = pa[r]; k < pa[r + 1]; ++k) a = fe_add<S>(a, fe_mul<S>(ld_fr<S>(ca + (size_t)k * S::EW), ld_fr<S>(w + (size_t)ia[k] * S::EW)));
        for (u32 k = pb[r]; k < pb[r + 1]; ++k) b = fe_add<S>(b, fe_mul<S>(ld_fr<S>(cb + (size_t)k * S::EW), ld_fr<S>(w + (size_t)ib[k] * S::EW)));
        st_fr<S>(fe_mul<S>(fe_mul<S>(a, b), c32), w + (size_t)ic[pc[r]] * S::EW);
    }
}
template <int EW> __global__ void k_gather_fr(const u32* __restrict__ src, const u32* __restrict__ idx, u32* __restrict__ dst, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)idx[i] * EW);
    uint4* d = reinterpret_cast<uint4*>(dst + i * EW);
#pragma unroll
    for (int k = 0; k < EW / 4; ++k) d[k] = s[k];
}
// L_j(tau) = (tau^n - 1)/n * w^j / (tau - w^j); consts = {w, tau, (tau^n - 1)/n}, all internal form
template <class S> __global__ void k_lagrange(u32* __restrict__ out, const u32* __restrict__ consts, size_t n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const Fe<S> w = ld_fr<S>(consts), tau = ld_fr<S>(consts + S::EW), zn = ld_fr<S>(consts + 2 * S::EW);
    const Fe<S> wj = fe_pow_u64<S>(w, (u64)j);
    const Fe<S> den = fe_inv<S>(fe_sub<S>(tau, wj));
    st_fr<S>(fe_mul<S>(fe_mul<S>(zn, wj), den), out + j * S::EW);
}
// column evaluation: out[i] = sum_k coef[k] * L[row[k]] over the entries of wire i (CSC of one R1CS matrix)
template <class S> __global__ void k_col_eval(const u32* __restrict__ ptr, const u32* __restrict__ row, const u32* __restrict__ coef,
                                  const u32* __restrict__ L, u32* __restrict__ out, u32 cols) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cols) return;
    Fe<S> acc = fe_zero<S>();
    for (u32 k = ptr[i]; k < ptr[i + 1]; ++k)
        acc = fe_add<S>(acc, fe_mul<S>(ld_fr<S>(coef + (size_t)k * S::EW), ld_fr<S>(L + (size_t)row[k] * S::EW)));
    st_fr<S>(acc, out + (size_t)i * S::EW);
}
// K_i = (beta * At_i + alpha * Bt_i + Ct_i) / delta for i >= n_public; consts = {alpha, beta, 1/delta}
template <class S> __global__ void k_key_k(const u32* __restrict__ at, const u32* __restrict__ bt, const u32* __restrict__ ct,
                               const u32* __restrict__ consts, u32* __restrict__ out, u32 first, u32 count) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t i = (size_t)first + t;
    const Fe<S> alpha = ld_fr<S>(consts), beta = ld_fr<S>(consts + S::EW), dinv = ld_fr<S>(consts + 2 * S::EW);
    Fe<S> v = fe_add<S>(fe_add<S>(fe_mul<S>(beta, ld_fr<S>(at + i * S::EW)), fe_mul<S>(alpha, ld_fr<S>(bt + i * S::EW))), ld_fr<S>(ct + i * S::EW));
    st_fr<S>(fe_mul<S>(v, dinv), out + (size_t)t * S::EW);
}
// Z'_i = zdt * tau^(bitrev(i)): gnark computes Z_i = (tau^n - 1)/delta * tau^i and bit-reverses the POINT array
template <class S> __global__ void k_key_z(u32* __restrict__ out, const u32* __restrict__ consts /* {tau, zdt} */, u32 log_n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const Fe<S> tau = ld_fr<S>(consts), zdt = ld_fr<S>(consts + S::EW);
    const u32 e = __brev(i) >> (32 - log_n);
    st_fr<S>(fe_mul<S>(zdt, fe_pow_u64<S>(tau, (u64)e)), out + (size_t)i * S::EW);
}
// fixed-base table: table[(w * 256 + d)] = d * 2^(8w) * G (affine internal), d in 1..255, w in 0..nwin-1 (nwin = scalar bytes)
template <class Cv> __global__ void k_fb_table(u32* __restrict__ table, const u32* __restrict__ gen, u32 nwin) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwin * 256) return;
    const u32 w = t >> 8, d = t & 255;
    typename Cv::X p = Cv::from_affine(ld_affine<Cv>(gen));
    for (u32 k = 0; k < 8 * w; ++k) p = Cv::dbl(p);
    typename Cv::A a = Cv::to_affine(Cv::mul_small(p, d));
    u32 wds[Cv::AW];
    Cv::store_affine_internal(a, wds);
    store_words16(table + (size_t)t * Cv::AW, wds, Cv::AW);
}
// out[i] = [k_i] G via the table (gnark's BatchScalarMultiplicationG1/G2 at Setup); scalars internal form
template <class Cv, class S> __global__ void k_fixed_base(const u32* __restrict__ table, const u32* __restrict__ scalars,
                                                 u32* __restrict__ out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u32 kw[S::EW];
    fe_pack<S>(fe_canonical<S>(fe_mul<S>(ld_fr<S>(scalars + i * S::EW), fe_small<S>(1))), kw);   // internal -> integer
    typename Cv::X acc = Cv::infinity();
    for (u32 w = 0; w < 4 * S::EW; ++w) {
        const u32 d = (kw[w >> 2] >> (8 * (w & 3))) & 255;
        if (d) acc = Cv::add_mixed(acc, ld_affine<Cv>(table + (size_t)(w * 256 + d) * Cv::AW));
    }
    typename Cv::A a = Cv::to_affine(acc);
    u32 wds[Cv::AW];
    Cv::store_affine_internal(a, wds);
    store_words16(out + i * Cv::AW, wds, Cv::AW);
}

// ------------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------------
struct SplitMix {
    u64 s;
    explicit SplitMix(u64 seed) : s(seed) {}
    u64 next() {
        s += 0x9E3779B97F4A7C15ULL;
        u64 z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    // canonical scalar as internal-form element: EW/2 stream words (the width of the field image) reduced mod r
    // (oracle SplitMix64.next_fr / curves_py.SuiteRng)
    template <class S> Fe<S> next_fr(const u32* rmod /* S::EW words */) {
        constexpr int EW = S::EW;
        u32 w[EW];
        for (int k = 0; k < EW / 2; ++k) { u64 v = next(); w[2 * k] = (u32)v; w[2 * k + 1] = (u32)(v >> 32); }
        auto geq = [&]() { for (int i = EW - 1; i >= 0; --i) { if (w[i] > rmod[i]) return true; if (w[i] < rmod[i]) return false; } return true; };
        while (geq()) { u64 br = 0; for (int i = 0; i < EW; ++i) { u64 d = (u64)w[i] - rmod[i] - br; w[i] = (u32)d; br = (d >> 32) & 1; } }
        return fe_from_canonical<S>(fe_unpack<S>(w));
    }
};
template <class S> static Fe<S> fr_u64(u64 v) { u32 w[S::EW] = {(u32)v, (u32)(v >> 32)}; return fe_from_canonical<S>(fe_unpack<S>(w)); }
template <class S> static void fr_pack_internal(const Fe<S>& v, u32* w) { u32 t[S::EW]; fe_pack<S>(fe_canonical<S>(v), t); memcpy(w, t, S::EW * 4); }
template <class S> static void fr_to_ext_words(const Fe<S>& v, u32* w) { u32 t[S::EW]; fe_to_ext_mont<S>(v, t); memcpy(w, t, S::EW * 4); }
template <class S> static void fr_canonical_words(const Fe<S>& v, u32* w) { u32 t[S::EW]; fe_pack<S>(fe_canonical<S>(fe_mul<S>(v, fe_small<S>(1))), t); memcpy(w, t, S::EW * 4); }
template <class S> static Fe<S> fr_from_ext_words(const u64* p) { u32 t[S::EW]; memcpy(t, p, S::EW * 4); return fe_from_ext_mont<S>(t); }

template <class T> static int32_t dev_upload(T** d, const std::vector<T>& h) {
    HIP_TRY(hipMalloc((void**)d, h.size() * sizeof(T) + 16));
    if (!h.empty()) HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return DVN_OK;
}

// ------------------------------------------------------------------------------------------------------
// synthetic R1CS (oracle/bn254_py.py SyntheticR1CS): rows (w[l1] + k*w[l2]) * w[r] = w[o], witness built forward
// ------------------------------------------------------------------------------------------------------
// n_committed > 0: one BSB22 commitment (oracle/bn254_py.py SyntheticR1CSCommitted): the commitment wire follows the public
// inputs, the first n_committed private wires are committed, only the last `tail` constraints read the commitment wire
// (its value, and those constraints' outputs, are placeholders in cs->witness until the prover completes them).
template <class SU> static int32_t r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out,
                                                  uint32_t n_committed = 0, uint32_t tail = 2) {
    typedef typename SU::FrC S;
    if (!out || n_public < 1) return fail(DVN_ERR_INVALID_ARG, "bad argument");
    if (n_committed && n_constraints <= tail) return fail(DVN_ERR_INVALID_ARG, "need more than %u constraints", tail);
    dvn_r1cs* cs = new (std::nothrow) dvn_r1cs();
    if (!cs) return fail(DVN_ERR_OOM, "host allocation failed");
    cs->suite = SU::SUITE;
    SplitMix rng(seed);
    const u32 n_free = 4;
    std::vector<Fe<S>> w;
    w.reserve((size_t)n_public + n_free + n_constraints);
    w.push_back(fe_one<S>());
    for (u32 i = 1; i < n_public; ++i) w.push_back(rng.template next_fr<S>(SU::r_words()));
    const u32 cw = n_committed ? n_public : 0xffffffffu;
    if (n_committed) {
        w.push_back(fe_zero<S>());                                   // the commitment wire closes the public section
        cs->commit_wire = n_public; cs->n_committed = n_committed; cs->tail = tail;
        if (n_public >= 4) cs->public_committed.push_back(3); else if (n_public >= 2) cs->public_committed.push_back(n_public - 1);
    }
    for (u32 i = 0; i < n_free + n_committed; ++i) w.push_back(rng.template next_fr<S>(SU::r_words()));
    for (int k = 0; k < 3; ++k) cs->m[k].ptr.push_back(0);
    auto push = [&](Csr& c, u32 wire, u64 coefv) { c.idx.push_back(wire); u32 t[S::EW]; fr_pack_internal<S>(fr_u64<S>(coefv), t); c.coef.insert(c.coef.end(), t, t + S::EW); };
    for (u32 j = 0; j < n_constraints; ++j) {
        const u64 mcur = w.size();
        u32 l1 = (u32)(rng.next() % mcur), l2 = (u32)(rng.next() % mcur);
        const u64 k = rng.next() % 7 + 1;
        u32 r_ = (u32)(rng.next() % mcur);
        const u32 o = (u32)mcur;
        if (n_committed) {
            if (l2 == cw) l2 = 0;
            if (r_ == cw) r_ = 0;
            l1 = (j >= n_constraints - tail) ? cw : (l1 == cw ? 0 : l1);
        }
        Fe<S> val = fe_mul<S>(fe_add<S>(w[l1], fe_mul<S>(fr_u64<S>(k), w[l2])), w[r_]);
        w.push_back(val);
        if (l1 != l2) { push(cs->m[0], l1, 1); push(cs->m[0], l2, k); } else push(cs->m[0], l1, 1 + k);
        push(cs->m[1], r_, 1);
        push(cs->m[2], o, 1);
        for (int q = 0; q < 3; ++q) cs->m[q].ptr.push_back((u32)cs->m[q].idx.size());
    }
    cs->n_wires = (u32)w.size(); cs->n_public = n_public + (n_committed ? 1 : 0); cs->n_constraints = n_constraints;
    cs->witness.resize((size_t)cs->n_wires * S::EW);
    for (size_t i = 0; i < w.size(); ++i) fr_to_ext_words<S>(w[i], cs->witness.data() + i * S::EW);
    *out = cs;
    return DVN_OK;
}
// ------------------------------------------------------------------------------------------------------
// Setup
// ------------------------------------------------------------------------------------------------------
template <class Cv, class S>
static int32_t fixed_base_points(dvn_ctx* ctx, const typename Cv::A& gen, const u32* d_scalars, size_t count, int curve, dvn_points** out) {
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = curve; p->n = count;
    expand_plan(count, S::BITS, (size_t)Cv::AW * 4, &p->expand_c, &p->nexp);
    p->suite = SuiteOf<Cv>::value;
    u32 *table = nullptr, *dgen = nullptr;
    u32 hg[Cv::AW];
    Cv::store_affine_internal(gen, hg);
    hipError_t e = hipMalloc((void**)&p->d, count ? count * Cv::AW * 4 * p->nexp : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&table, (size_t)4 * S::EW * 256 * Cv::AW * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dgen, sizeof(hg));
    if (e == hipSuccess) e = hipMemcpy(dgen, hg, sizeof(hg), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_fb_table<Cv><<<4 * S::EW * 256 / 64, 64, 0, ctx->stream>>>(table, dgen, 4 * S::EW);
        if (count) k_fixed_base<Cv, S><<<(unsigned)((count + 63) / 64), 64, 0, ctx->stream>>>(table, d_scalars, p->d, count);
        if (count && p->expand_c) k_points_expand<Cv><<<(unsigned)((count + 63) / 64), 64, 0, ctx->stream>>>(p->d, count, p->expand_c, p->nexp);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (table) (void)hipFree(table);
    if (dgen) (void)hipFree(dgen);
    if (e != hipSuccess) { if (p->d) (void)hipFree(p->d); delete p; (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "fixed-base multiplication failed: %s", hipGetErrorString(e)); }
    *out = p;
    return DVN_OK;
}

// groth16.Setup for the sparse R1CS with toxic waste drawn from SplitMix64(seed) (tau, alpha, beta, gamma, delta in
// that order, 0 replaced by 1): deterministic so the trapdoor identity can be checked by the oracle.
template <class SU> static int32_t groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out) {
    typedef typename SU::FrC S;
    typedef typename SU::G1T G1;
    typedef typename SU::G2T G2;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    constexpr u32 EW = S::EW;                  // u32 words of a scalar image
    constexpr size_t EB = (size_t)EW * 4;
    TRY(enter(ctx));
    if (!cs || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (cs->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "constraint system belongs to another curve");
    if (cs->n_constraints < 2) return fail(DVN_ERR_INVALID_ARG, "need at least 2 constraints");
    u32 log_n = 1;
    while (((size_t)1 << log_n) < cs->n_constraints) ++log_n;
    const size_t n = (size_t)1 << log_n, m = cs->n_wires;
    dvn_groth16_pk* pk = new (std::nothrow) dvn_groth16_pk();
    if (!pk) return fail(DVN_ERR_OOM, "host allocation failed");
    pk->suite = SU::SUITE; pk->device = ctx->device;
    pk->scalar_bytes = S::EW * 4; pk->g1_bytes = G1::AW * 4; pk->g2_bytes = G2::AW * 4;
    pk->log_n = log_n; pk->n = n; pk->m = m; pk->n_public = cs->n_public; pk->n_constraints = cs->n_constraints;
    int32_t rc = SU::domain_create(ctx, log_n, &pk->dom);
    if (rc) { delete pk; return rc; }
    std::unique_lock<std::mutex> lk(ctx->mu);
    SplitMix rng(seed);
    Fe<S> tw[6];
    for (int i = 0; i < (cs->n_committed ? 6 : 5); ++i) { tw[i] = rng.template next_fr<S>(SU::r_words()); if (fe_is_zero<S>(tw[i])) tw[i] = fe_one<S>(); }
    pk->k_first = (size_t)cs->n_public + cs->n_committed;
    pk->commit_wire = cs->commit_wire; pk->n_committed = cs->n_committed; pk->tail = cs->tail; pk->public_committed = cs->public_committed;
    const Fe<S> tau = tw[0], alpha = tw[1], beta = tw[2], delta = tw[4];
    const Fe<S> one = fe_one<S>(), dinv = fe_inv<S>(delta);
    const Fe<S> ztau = fe_sub<S>(fe_pow_u64<S>(tau, n), one);
    const Fe<S> omega = fr_root_of_unity<S>(log_n);          // ntt_driver.h: the domain's generator

    u32 *d_L = nullptr, *d_t[3] = {nullptr, nullptr, nullptr}, *d_k = nullptr, *d_z = nullptr, *d_consts = nullptr;
    u32 *d_cptr = nullptr, *d_crow = nullptr, *d_ccoef = nullptr, *d_cmp = nullptr;
    auto cleanup = [&]() {
        u32* ps[] = {d_L, d_t[0], d_t[1], d_t[2], d_k, d_z, d_consts, d_cptr, d_crow, d_ccoef, d_cmp};
        for (u32* p : ps) if (p) (void)hipFree(p);
    };
    auto body = [&]() -> int32_t {
        hipStream_t st = ctx->stream;
        HIP_TRY(hipMalloc((void**)&d_L, n * EB));
        HIP_TRY(hipMalloc((void**)&d_consts, 8 * EB));
        for (int q = 0; q < 3; ++q) HIP_TRY(hipMalloc((void**)&d_t[q], m * EB));
        u32 hc[8 * EW];
        fr_pack_internal<S>(omega, hc); fr_pack_internal<S>(tau, hc + EW);
        fr_pack_internal<S>(fe_mul<S>(ztau, fe_inv<S>(fr_u64<S>(n))), hc + 2 * EW);
        fr_pack_internal<S>(alpha, hc + 3 * EW); fr_pack_internal<S>(beta, hc + 4 * EW); fr_pack_internal<S>(dinv, hc + 5 * EW);
        fr_pack_internal<S>(tau, hc + 6 * EW); fr_pack_internal<S>(fe_mul<S>(ztau, dinv), hc + 7 * EW);
        HIP_TRY(hipMemcpy(d_consts, hc, sizeof(hc), hipMemcpyHostToDevice));
        k_lagrange<S><<<(unsigned)((n + 63) / 64), 64, 0, st>>>(d_L, d_consts, n);
        HIP_TRY(hipGetLastError());
        // column (CSC) form of each matrix, evaluated at tau
        for (int q = 0; q < 3; ++q) {
            const Csr& r = cs->m[q];
            std::vector<u32> cptr(m + 1, 0), crow(r.idx.size()), ccoef(r.idx.size() * EW);
            for (u32 wi : r.idx) cptr[wi + 1]++;
            for (size_t i = 0; i < m; ++i) cptr[i + 1] += cptr[i];
            std::vector<u32> cur(cptr.begin(), cptr.end() - 1);
            for (u32 row = 0; row < cs->n_constraints; ++row)
                for (u32 k = r.ptr[row]; k < r.ptr[row + 1]; ++k) {
                    const u32 pos = cur[r.idx[k]]++;
                    crow[pos] = row; memcpy(&ccoef[(size_t)pos * EW], &r.coef[(size_t)k * EW], EB);
                }
            TRY(dev_upload(&d_cptr, cptr)); TRY(dev_upload(&d_crow, crow)); TRY(dev_upload(&d_ccoef, ccoef));
            k_col_eval<S><<<(unsigned)((m + 63) / 64), 64, 0, st>>>(d_cptr, d_crow, d_ccoef, d_L, d_t[q], (u32)m);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(d_cptr); (void)hipFree(d_crow); (void)hipFree(d_ccoef); d_cptr = d_crow = d_ccoef = nullptr;
            // row-major copy for Prove's matvec
            TRY(dev_upload(&pk->d_ptr[q], r.ptr)); TRY(dev_upload(&pk->d_idx[q], r.idx)); TRY(dev_upload(&pk->d_coef[q], r.coef));
        }
        // K (private wires) and Z (bit-reversed)
        const size_t nk = m - pk->k_first;
        HIP_TRY(hipMalloc((void**)&d_k, (nk ? nk : 1) * EB));
        HIP_TRY(hipMalloc((void**)&d_z, n * EB));
        if (nk) k_key_k<S><<<(unsigned)((nk + 63) / 64), 64, 0, st>>>(d_t[0], d_t[1], d_t[2], d_consts + 3 * EW, d_k, (u32)pk->k_first, (u32)nk);
        k_key_z<S><<<(unsigned)((n + 63) / 64), 64, 0, st>>>(d_z, d_consts + 6 * EW, log_n);
        HIP_TRY(hipGetLastError());
        // infinity filtering of A and B (gnark's InfinityA / InfinityB)
        std::vector<u32> hA(m * EW), hB(m * EW);
        HIP_TRY(hipMemcpyAsync(hA.data(), d_t[0], m * EB, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hB.data(), d_t[1], m * EB, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<u32> idxA, idxB;
        auto nonzero = [](const u32* p) { u32 z = 0; for (u32 k = 0; k < EW; ++k) z |= p[k]; return z != 0; };
        for (u32 i = 0; i < m; ++i) { if (nonzero(&hA[(size_t)i * EW])) idxA.push_back(i); if (nonzero(&hB[(size_t)i * EW])) idxB.push_back(i); }
        pk->nA = idxA.size(); pk->nB = idxB.size();
        TRY(dev_upload(&pk->d_idxA, idxA)); TRY(dev_upload(&pk->d_idxB, idxB));
        const size_t cmax = std::max(pk->nA, pk->nB);
        HIP_TRY(hipMalloc((void**)&d_cmp, (cmax ? cmax : 1) * EB));
        const typename G1::A g1 = SU::g1_gen();
        const typename G2::A g2 = SU::g2_gen();
        if (pk->nA) k_gather_fr<S::EW><<<(unsigned)((pk->nA + 255) / 256), 256, 0, st>>>(d_t[0], pk->d_idxA, d_cmp, pk->nA);
        HIP_TRY(hipGetLastError());
        TRY((fixed_base_points<G1, S>(ctx, g1, d_cmp, pk->nA, 1, &pk->A)));
        if (pk->nB) k_gather_fr<S::EW><<<(unsigned)((pk->nB + 255) / 256), 256, 0, st>>>(d_t[1], pk->d_idxB, d_cmp, pk->nB);
        HIP_TRY(hipGetLastError());
        TRY((fixed_base_points<G1, S>(ctx, g1, d_cmp, pk->nB, 1, &pk->B)));
        TRY((fixed_base_points<G2, S>(ctx, g2, d_cmp, pk->nB, 2, &pk->B2)));
        TRY((fixed_base_points<G1, S>(ctx, g1, d_k, nk, 1, &pk->K)));
        if (cs->n_committed) {
            // commitment key (gnark groth16 Setup + pedersen.Setup): Basis_i = [(beta A_i + alpha B_i + C_i)(tau) / gamma]G1 for the
            // committed wires -- the commitment is added to the gamma-paired public-input accumulator by the verifier
            // (/root/reference/config/statetransition_vkey.sol:412-416) -- and BasisExpSigma_i = sigma * Basis_i
            const Fe<S> ginv = fe_inv<S>(tw[3]);
            u32 hb[6 * EW];
            fr_pack_internal<S>(alpha, hb); fr_pack_internal<S>(beta, hb + EW); fr_pack_internal<S>(ginv, hb + 2 * EW);
            fr_pack_internal<S>(alpha, hb + 3 * EW); fr_pack_internal<S>(beta, hb + 4 * EW); fr_pack_internal<S>(fe_mul<S>(ginv, tw[5]), hb + 5 * EW);
            u32* d_bc = nullptr;
            HIP_TRY(hipMalloc((void**)&d_bc, sizeof(hb) + (size_t)cs->n_committed * EB));
            u32* d_bs = d_bc + 6 * EW;
            hipError_t e = hipMemcpy(d_bc, hb, sizeof(hb), hipMemcpyHostToDevice);
            int32_t rc2 = e == hipSuccess ? DVN_OK : fail(DVN_ERR_DEVICE, "commitment key upload failed: %s", hipGetErrorString(e));
            for (int which = 0; which < 2 && rc2 == DVN_OK; ++which) {
                k_key_k<S><<<(cs->n_committed + 63) / 64, 64, 0, st>>>(d_t[0], d_t[1], d_t[2], d_bc + 3 * which * EW, d_bs, cs->n_public, cs->n_committed);
                rc2 = fixed_base_points<G1, S>(ctx, g1, d_bs, cs->n_committed, 1, which ? &pk->BasisSigma : &pk->Basis);
            }
            (void)hipFree(d_bc);
            TRY(rc2);
        }
        TRY((fixed_base_points<G1, S>(ctx, g1, d_z, n, 1, &pk->Z)));
        // per-proof buffers
        HIP_TRY(hipMalloc((void**)&pk->d_w, m * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_abc, 3 * n * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wA, (pk->nA ? pk->nA : 1) * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wB, (pk->nB ? pk->nB : 1) * EB));
        // G1/G2 constants of the key
        u32 kw[EW], img1[G1::AW], img2[G2::AW];
        G1::store_affine_internal(g1, img1); G2::store_affine_internal(g2, img2);
        const typename H1::X hg1 = H1::load_device_affine(img1);
        const typename H2::X hg2 = H2::load_device_affine(img2);
        HostKey<SU>* hk = new HostKey<SU>();
        pk->hk = hk; pk->hk_free = [](void* p) { delete (HostKey<SU>*)p; };
        fr_canonical_words<S>(alpha, kw); hk->alpha = H1::normalize(host_mul<H1, S>(hg1, kw));
        fr_canonical_words<S>(beta, kw); hk->beta = H1::normalize(host_mul<H1, S>(hg1, kw));
        hk->beta2 = H2::normalize(host_mul<H2, S>(hg2, kw));
        fr_canonical_words<S>(delta, kw); hk->delta = H1::normalize(host_mul<H1, S>(hg1, kw));
        hk->delta2 = H2::normalize(host_mul<H2, S>(hg2, kw));
        // verifying key (groth16.Setup returns (pk, vk), /root/reference/prover/setup.go:15-28): alpha, beta, gamma, delta and
        // IC_i = [(beta A_i + alpha B_i + C_i)(tau) / gamma]G1 over the public section; with a commitment the Pedersen pair
        // G, GSigmaNeg = -sigma G (statetransition_vkey.sol:62-114 holds exactly these constants, beta/gamma/delta negated)
        {
            constexpr size_t A1 = G1::AW / 2, A2 = G2::AW / 2;
            auto put1 = [&](const typename H1::X& pnt) { uint64_t j[3 * A1 / 2 + 1]; H1::write_jac(pnt, j); if (H1::is_inf(pnt)) memset(j, 0, sizeof(j)); pk->vk_words.insert(pk->vk_words.end(), j, j + A1); };
            auto put2 = [&](const typename H2::X& pnt) { uint64_t j[3 * A2 / 2 + 1]; H2::write_jac(pnt, j); if (H2::is_inf(pnt)) memset(j, 0, sizeof(j)); pk->vk_words.insert(pk->vk_words.end(), j, j + A2); };
            put1(hk->alpha); put2(hk->beta2);
            fr_canonical_words<S>(tw[3], kw); put2(host_mul<H2, S>(hg2, kw));
            put2(hk->delta2);
            const Fe<S> ginv = fe_inv<S>(tw[3]);
            u32 hv[3 * EW];
            fr_pack_internal<S>(alpha, hv); fr_pack_internal<S>(beta, hv + EW); fr_pack_internal<S>(ginv, hv + 2 * EW);
            u32* d_v = nullptr;
            HIP_TRY(hipMalloc((void**)&d_v, sizeof(hv) + (size_t)cs->n_public * EB));
            hipError_t e = hipMemcpy(d_v, hv, sizeof(hv), hipMemcpyHostToDevice);
            dvn_points* ic = nullptr;
            int32_t rc3 = e == hipSuccess ? DVN_OK : fail(DVN_ERR_DEVICE, "vk upload failed: %s", hipGetErrorString(e));
            if (rc3 == DVN_OK) {
                k_key_k<S><<<(cs->n_public + 63) / 64, 64, 0, st>>>(d_t[0], d_t[1], d_t[2], d_v, d_v + 3 * EW, 0, cs->n_public);
                rc3 = fixed_base_points<G1, S>(ctx, g1, d_v + 3 * EW, cs->n_public, 1, &ic);
            }
            (void)hipFree(d_v);
            TRY(rc3);
            const size_t at = pk->vk_words.size();
            pk->vk_words.resize(at + (size_t)cs->n_public * A1);
            rc3 = points_download<G1>(ctx, ic, 0, cs->n_public, pk->vk_words.data() + at);
            (void)hipFree(ic->d); delete ic;
            TRY(rc3);
            if (cs->n_committed) {
                put2(hg2);
                fr_canonical_words<S>(fe_neg<S>(tw[5]), kw); put2(host_mul<H2, S>(hg2, kw));
            }
        }
        return DVN_OK;
    };
    rc = body();
    cleanup();
    lk.unlock();
    if (rc != DVN_OK) { std::string keep = tl_error; (void)dvn_groth16_pk_free(ctx, pk); tl_error = keep; return rc; }
    *out = pk;
    return DVN_OK;
}

// ------------------------------------------------------------------------------------------------------
// Prove
// ------------------------------------------------------------------------------------------------------
#define TRY_HOLD(expr) do { int32_t _rc = (expr); if (_rc != DVN_OK) { ctx->dep_hold = false; return _rc; } } while (0)
template <class SU> struct MsmResults { typename SU::H1::X mA, mB, mK, mZ; typename SU::H2::X mB2; };

// Device part of a proof.  world == 1: the five full MSMs.  world > 1 (sharded proving):
// this rank's slice of the ROWS of every (expanded) key set -- the partial results of all ranks add up to the MSMs; the
// "solve" and computeH are replicated on every rank.  Caller holds ctx->mu.
template <class SU> static int32_t groth16_device(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                                  u32 rank, u32 world, MsmResults<SU>* out,
                                                  const uint64_t* const* abc_host = nullptr, size_t abc_len = 0) {
    typedef typename SU::FrC S;
    constexpr u32 EW = S::EW;
    constexpr size_t EB = (size_t)EW * 4;
    hipStream_t st = ctx->stream;
    const size_t n = pk->n, m = pk->m;
    hipEvent_t e0, e1, e2;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2));
    auto submit_g1 = [&](const dvn_points* p, size_t cnt, const u32* sc, MsmLane** l) -> int32_t {
        if (world == 1) return SU::msm_submit_g1(ctx, p, 0, cnt, sc, l);
        return SU::msm_share_submit_g1(ctx, p, 0, cnt, sc, rank, world, l);
    };
    auto submit_g2 = [&](const dvn_points* p, size_t cnt, const u32* sc, MsmLane** l) -> int32_t {
        if (world == 1) return SU::msm_submit_g2(ctx, p, 0, cnt, sc, l);
        return SU::msm_share_submit_g2(ctx, p, 0, cnt, sc, rank, world, l);
    };
    // every lane this proof acquires is tracked here: if a later submit, computeH or a collect fails, the lanes still in
    // flight are drained and released, so one failed proof never leaves the context with busy lanes
    MsmLane* lanes[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // B2, A, B1, K, Z
    pk->h2d_bytes = 0; pk->last_rank = rank; pk->last_world = world;
    auto body = [&]() -> int32_t {
        MsmLane *&l2 = lanes[0], *&lA = lanes[1], *&lB = lanes[2], *&lK = lanes[3], *&lZ = lanes[4];
        static const bool trace_host = getenv("DVN_TRACE_HOST") != nullptr;
        const auto th0 = std::chrono::steady_clock::now();
        auto tmark = [&](const char* what) { if (trace_host) fprintf(stderr, "[dvn host] %8.3f ms %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count(), what); };
        HIP_TRY(hipEventRecord(e0, st));
        if (!(flags & 2)) {    // flags & 2: the (completed) witness already sits in pk->d_w (committed proofs)
            RoctxRange rr("dvn:witness_copy");
            if (flags & 1) HIP_TRY(hipMemcpyAsync(pk->d_w, witness, m * EB, hipMemcpyDeviceToDevice, st));
            else TRY(dvn_h2d(ctx, pk->d_w, witness, m * EB, st));
            if (!(flags & 1)) pk->h2d_bytes += m * EB;
        }
        // (4) filter the wire vector by InfinityA / InfinityB
        {
            RoctxRange rr("dvn:wire_filter");
            if (pk->nA) k_gather_fr<S::EW><<<(unsigned)((pk->nA + 255) / 256), 256, 0, st>>>(pk->d_w, pk->d_idxA, pk->d_wA, pk->nA);
            if (pk->nB) k_gather_fr<S::EW><<<(unsigned)((pk->nB + 255) / 256), 256, 0, st>>>(pk->d_w, pk->d_idxB, pk->d_wB, pk->nB);
            // imported key: K's wires are an index list (private minus committed)
            if (pk->d_idxK && pk->nK) k_gather_fr<S::EW><<<(unsigned)((pk->nK + 255) / 256), 256, 0, st>>>(pk->d_w, pk->d_idxK, pk->d_wK, pk->nK);
            HIP_TRY(hipGetLastError());
        }
        // (6) the four MSMs that need only the wire values go to four lanes at once (a lane waits for what the main stream
        // holds at submit time, i.e. the gathers): their sorts and latency-bound bucket reductions overlap each other ...
        static const int solve_first = [] { const char* e = getenv("DVN_G16_SOLVE_FIRST"); return e ? atoi(e) : 0; }();
        auto queue_solve = [&]() -> int32_t {
            // ... while the main stream runs (1) the "solve" a = A w, b = B w, c = C w (zero padded to the domain size) and
            // (3) h = computeH(a, b, c), which stays on the device as the scalar vector of the Z MSM
            if (abc_host) {                                    // the solver's (a, b, c) from host memory: their upload runs under the wire-only MSMs
                RoctxRange rr("dvn:abc_copy");
                for (int q = 0; q < 3; ++q) {
                    // plain copies: these run beside the accumulate kernels of the wire-only MSMs, where the staged path (dvn_h2d) measured
                    // 0.7 ms SLOWER per proof than the runtime's own pageable copy; it pays only while the GPU is otherwise idle
                    if (abc_len) HIP_TRY(hipMemcpyAsync(pk->d_abc + (size_t)q * n * EW, abc_host[q], abc_len * EB, hipMemcpyHostToDevice, st));
                    TRY(dvn_zero_async(pk->d_abc + ((size_t)q * n + abc_len) * EW, (n - abc_len) * EB, st));      // zero padding to the domain size
                }
                pk->h2d_bytes += 3 * abc_len * EB;
            } else if (!(flags & 4)) {                         // flags & 4: (a, b, c) already sit zero-padded in pk->d_abc
                RoctxRange rr("dvn:r1cs_matvec");
                for (int q = 0; q < 3; ++q)
                    k_r1cs_matvec<S><<<(unsigned)((n + 127) / 128), 128, 0, st>>>(pk->d_ptr[q], pk->d_idx[q], pk->d_coef[q], pk->d_w,
                                                                                  pk->d_abc + (size_t)q * n * EW, (u32)pk->n_constraints, (u32)n);
                HIP_TRY(hipGetLastError());
            }
            { RoctxRange rr("dvn:compute_h"); TRY(SU::compute_h(ctx, pk->dom, pk->d_abc, pk->d_abc + n * EW, pk->d_abc + 2 * n * EW)); }
            HIP_TRY(hipEventRecord(e1, st));
            return DVN_OK;
        };
        if (solve_first) {
            // the solve and computeH go to the main stream FIRST; the lanes wait for the gathers only (ev_dep recorded here, held)
            HIP_TRY(hipEventRecord(ctx->ev_dep, st));
            ctx->dep_hold = true;
            const int32_t rcs = queue_solve();
            if (rcs != DVN_OK) { ctx->dep_hold = false; return rcs; }
        }
        tmark("gathers queued");
        { RoctxRange rr("dvn:msm_submit:G2.B"); TRY_HOLD(submit_g2(pk->B2, pk->nB, pk->d_wB, &l2)); }
        tmark("G2.B submitted");
        { RoctxRange rr("dvn:msm_submit:G1.A"); TRY_HOLD(submit_g1(pk->A, pk->nA, pk->d_wA, &lA)); }
        tmark("G1.A submitted");
        { RoctxRange rr("dvn:msm_submit:G1.B"); TRY_HOLD(submit_g1(pk->B, pk->nB, pk->d_wB, &lB)); }
        tmark("G1.B submitted");
        size_t nK = 0;
        {
            RoctxRange rr("dvn:msm_submit:G1.K");
            if (pk->d_idxK) {
                nK = pk->nK;
                TRY_HOLD(submit_g1(pk->K, pk->nK, pk->d_wK, &lK));
            } else { nK = m - pk->k_first; TRY_HOLD(submit_g1(pk->K, nK, pk->d_w + pk->k_first * EW, &lK)); }
        }
        ctx->dep_hold = false;
        if (!solve_first) TRY(queue_solve());
        tmark("MSMs and computeH queued");
        // Z takes the first lane that frees up: its submit makes the lane wait for computeH
        RoctxRange rc_("dvn:msm_collect");
        TRY(SU::msm_collect_g1(ctx, lA, &out->mA)); pk->msm_ms[0] = lA->ms[0]; pk->msm_pairs[0] = pk->nA; lA = nullptr;
        tmark("A collected");
        { RoctxRange rr("dvn:msm_submit:G1.Z"); TRY(submit_g1(pk->Z, n - 1, pk->d_abc, &lZ)); }                // h[:n-1] against Z[:n-1]
        tmark("Z submitted");
        TRY(SU::msm_collect_g1(ctx, lB, &out->mB)); pk->msm_ms[1] = lB->ms[0]; pk->msm_pairs[1] = pk->nB; lB = nullptr;
        TRY(SU::msm_collect_g1(ctx, lK, &out->mK)); pk->msm_ms[2] = lK->ms[0]; pk->msm_pairs[2] = nK; lK = nullptr;
        TRY(SU::msm_collect_g1(ctx, lZ, &out->mZ)); pk->msm_ms[3] = lZ->ms[0]; pk->msm_pairs[3] = n - 1; lZ = nullptr;
        TRY(SU::msm_collect_g2(ctx, l2, &out->mB2)); pk->msm_ms[4] = l2->ms[0]; pk->msm_pairs[4] = pk->nB; l2 = nullptr;
        HIP_TRY(hipEventRecord(e2, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipEventElapsedTime(&pk->ms_h, e0, e1));
        HIP_TRY(hipEventElapsedTime(&pk->ms_msm, e1, e2));
        HIP_TRY(hipEventElapsedTime(&pk->ms_total, e0, e2));
        return DVN_OK;
    };
    int32_t rc = body();
    if (rc != DVN_OK) {
        std::string keep = tl_error;
        (void)hipStreamSynchronize(st);                   // the caller's host buffers may still be the source of queued copies
        for (MsmLane* L : lanes)
            if (L && L->busy) { (void)hipStreamSynchronize(L->stream); L->busy = false; }
        (void)hipGetLastError();
        tl_error = keep;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return rc;
}

// (5)(7) blinding and assembly on the host: O(1) group operations.  The four scalar multiplications that need only (r, s) and
// the key start on threads right away (Blind), so they run under the GPU work of the same proof.
template <class SU> struct Blind {
    typedef typename SU::FrC S;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    u32 rw[S::EW], sw[S::EW], krw[S::EW];
    std::future<typename H1::X> dr, ds, dk;
    std::future<typename H2::X> d2s;
    Blind(const HostKey<SU>& hk, const uint64_t* r_mont, const uint64_t* s_mont) {
        const Fe<S> r = fr_from_ext_words<S>(r_mont), s = fr_from_ext_words<S>(s_mont);
        fr_canonical_words<S>(r, rw); fr_canonical_words<S>(s, sw);
        fr_canonical_words<S>(fe_neg<S>(fe_mul<S>(r, s)), krw);
        dr = std::async(std::launch::async, [this, &hk] { return host_mul<H1, S>(hk.delta, rw); });
        ds = std::async(std::launch::async, [this, &hk] { return host_mul<H1, S>(hk.delta, sw); });
        d2s = std::async(std::launch::async, [this, &hk] { return host_mul<H2, S>(hk.delta2, sw); });
        dk = std::async(std::launch::async, [this, &hk] { return host_mul<H1, S>(hk.delta, krw); });
    }
};
template <class SU> static void groth16_assemble(const HostKey<SU>& hk, const MsmResults<SU>& R, Blind<SU>& bl, uint64_t* proof_out) {
    typedef typename SU::FrC S;
    typedef typename SU::G1T G1;
    typedef typename SU::G2T G2;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    const typename H1::X ar = H1::normalize(H1::add(H1::add(R.mA, hk.alpha), bl.dr.get()));
    const typename H1::X bs1 = H1::normalize(H1::add(H1::add(R.mB, hk.beta), bl.ds.get()));
    auto a_ars = std::async(std::launch::async, [&] { return host_mul<H1, S>(ar, bl.sw); });
    auto a_bsr = std::async(std::launch::async, [&] { return host_mul<H1, S>(bs1, bl.rw); });
    const typename H2::X bs = H2::add(H2::add(R.mB2, hk.beta2), bl.d2s.get());
    typename H1::X krs = H1::add(R.mK, R.mZ);
    krs = H1::add(krs, bl.dk.get());
    krs = H1::add(krs, a_ars.get());
    krs = H1::add(krs, a_bsr.get());
    constexpr size_t A1 = G1::AW / 2, A2 = G2::AW / 2;            // uint64 words of an affine image
    uint64_t j1[3 * A1 / 2 + 1], j2[3 * A2 / 2 + 1];
    H1::write_jac(ar, j1); memcpy(proof_out, j1, A1 * 8);
    H2::write_jac(bs, j2); memcpy(proof_out + A1, j2, A2 * 8);
    H1::write_jac(krs, j1); memcpy(proof_out + A1 + A2, j1, A1 * 8);
    if (H1::is_inf(ar)) memset(proof_out, 0, A1 * 8);             // gnark's affine infinity is (0, 0)
    if (H2::is_inf(bs)) memset(proof_out + A1, 0, A2 * 8);
    if (H1::is_inf(krs)) memset(proof_out + A1 + A2, 0, A1 * 8);
}

// witness: m x EW/2 uint64 gnark Montgomery words (host, or device with flags & 1); r_mont/s_mont: the blinding
// factors gnark draws from crypto/rand, injected so results are reproducible.  proof_out: Ar | Bs | Krs as affine
// Montgomery images exactly as gnark's Proof holds them (G1Affine, G2Affine, G1Affine of the curve).
template <class SU> static int32_t groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                                 const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) {
    TRY(enter(ctx));
    if (!pk || !witness || !r_mont || !s_mont || !proof_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pk->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "proving key belongs to another curve");
    if (pk->imported) return fail(DVN_ERR_INVALID_ARG, "imported key: it has no constraint system, prove from the solver's output (prove_solved)");
    if (pk->n_committed) return fail(DVN_ERR_INVALID_ARG, "the key carries a BSB22 commitment: the plain prover would leave the commitment wire unset and the proof could not verify; use dvn_groth16_prove_committed");
    const HostKey<SU>& hk = *(const HostKey<SU>*)pk->hk;
    std::lock_guard<std::mutex> lk(ctx->mu);
    Blind<SU> bl(hk, r_mont, s_mont);
    MsmResults<SU> R;
    TRY(groth16_device<SU>(ctx, pk, witness, flags & 1u, 0, 1, &R));
    groth16_assemble<SU>(hk, R, bl, proof_out);
    return DVN_OK;
}

// ---- keys that come from gnark: the in-memory arrays of a groth16.ProvingKey, handed over once ("hold device
// handles in a side table keyed by PK pointer") -- the path a node takes with the proving keys it loads at
// /root/reference/circuits/artifacts.go:391-406.  All point arrays are gnark affine Montgomery images; idxA / idxB / idxK list,
// in order, the wires whose scalars go with A / B (= B2) / K (gnark drops the points at infinity of A and B -- InfinityA/B --
// and keeps K for the private, non-committed wires only).
struct Groth16Import {
    u32 log_n; size_t m;
    const uint64_t *A, *B, *K, *Z, *B2; size_t nA, nB, nK, nZ;
    const u32 *idxA, *idxB, *idxK;
    const uint64_t *alpha1, *beta1, *delta1, *beta2, *delta2;
};
template <class SU> static int32_t groth16_pk_import(dvn_ctx* ctx, const Groth16Import& d, dvn_groth16_pk** out) {
    typedef typename SU::FrC S;
    typedef typename SU::G1T G1;
    typedef typename SU::G2T G2;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    constexpr size_t EB = (size_t)S::EW * 4;
    TRY(enter(ctx));
    if (!out || !d.A || !d.B || !d.K || !d.Z || !d.B2 || !d.idxA || !d.idxB || !d.idxK || !d.alpha1 || !d.beta1 || !d.delta1 || !d.beta2 || !d.delta2)
        return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const size_t n = (size_t)1 << d.log_n;
    if (d.log_n < 1 || d.log_n > 27 || d.nZ + 1 < n || d.m == 0) return fail(DVN_ERR_INVALID_ARG, "bad sizes: Z must hold at least n - 1 points");
    for (size_t i = 0; i < d.nA; ++i) if (d.idxA[i] >= d.m) return fail(DVN_ERR_INVALID_ARG, "idxA[%zu] outside the wire vector", i);
    for (size_t i = 0; i < d.nB; ++i) if (d.idxB[i] >= d.m) return fail(DVN_ERR_INVALID_ARG, "idxB[%zu] outside the wire vector", i);
    for (size_t i = 0; i < d.nK; ++i) if (d.idxK[i] >= d.m) return fail(DVN_ERR_INVALID_ARG, "idxK[%zu] outside the wire vector", i);
    dvn_groth16_pk* pk = new (std::nothrow) dvn_groth16_pk();
    if (!pk) return fail(DVN_ERR_OOM, "host allocation failed");
    pk->suite = SU::SUITE; pk->imported = true; pk->device = ctx->device;
    pk->scalar_bytes = S::EW * 4; pk->g1_bytes = G1::AW * 4; pk->g2_bytes = G2::AW * 4;
    pk->log_n = d.log_n; pk->n = n; pk->m = d.m; pk->n_public = 0; pk->n_constraints = n; pk->nA = d.nA; pk->nB = d.nB; pk->nK = d.nK; pk->k_first = d.m - d.nK;
    auto body = [&]() -> int32_t {
        TRY(SU::domain_create(ctx, d.log_n, &pk->dom));
        TRY((points_upload<G1>(ctx, d.A, d.nA, 1, &pk->A)));
        TRY((points_upload<G1>(ctx, d.B, d.nB, 1, &pk->B)));
        TRY((points_upload<G1>(ctx, d.K, d.nK, 1, &pk->K)));
        TRY((points_upload<G1>(ctx, d.Z, d.nZ, 1, &pk->Z)));
        TRY((points_upload<G2>(ctx, d.B2, d.nB, 2, &pk->B2)));
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto up = [&](u32** dst, const u32* src, size_t cnt) -> int32_t {
            HIP_TRY(hipMalloc((void**)dst, (cnt ? cnt : 1) * 4));
            if (cnt) HIP_TRY(hipMemcpy(*dst, src, cnt * 4, hipMemcpyHostToDevice));
            return DVN_OK;
        };
        TRY(up(&pk->d_idxA, d.idxA, d.nA)); TRY(up(&pk->d_idxB, d.idxB, d.nB)); TRY(up(&pk->d_idxK, d.idxK, d.nK));
        HIP_TRY(hipMalloc((void**)&pk->d_w, d.m * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_abc, 3 * n * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wA, (d.nA ? d.nA : 1) * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wB, (d.nB ? d.nB : 1) * EB));
        HIP_TRY(hipMalloc((void**)&pk->d_wK, (d.nK ? d.nK : 1) * EB));
        HostKey<SU>* hk = new HostKey<SU>();
        pk->hk = hk; pk->hk_free = [](void* p) { delete (HostKey<SU>*)p; };
        auto ld1 = [](const uint64_t* w) { u32 img[G1::AW]; G1::store_affine_internal(G1::affine_from_ext((const u32*)w), img); return H1::load_device_affine(img); };
        auto ld2 = [](const uint64_t* w) { u32 img[G2::AW]; G2::store_affine_internal(G2::affine_from_ext((const u32*)w), img); return H2::load_device_affine(img); };
        hk->alpha = ld1(d.alpha1); hk->beta = ld1(d.beta1); hk->delta = ld1(d.delta1); hk->beta2 = ld2(d.beta2); hk->delta2 = ld2(d.delta2);
        return DVN_OK;
    };
    int32_t rc = body();
    if (rc != DVN_OK) { std::string keep = tl_error; (void)dvn_groth16_pk_free(ctx, pk); tl_error = keep; return rc; }
    *out = pk;
    return DVN_OK;
}
template <class SU> static int32_t groth16_pk_import_api(dvn_ctx* ctx, const dvn_groth16_key_arrays* k, dvn_groth16_pk** out) {
    if (!k) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    Groth16Import d{k->log_n, (size_t)k->n_wires, k->A, k->B, k->K, k->Z, k->B2, (size_t)k->nA, (size_t)k->nB, (size_t)k->nK, (size_t)k->nZ,
                    k->idxA, k->idxB, k->idxK, k->alpha_g1, k->beta_g1, k->delta_g1, k->beta_g2, k->delta_g2};
    return groth16_pk_import<SU>(ctx, d, out);
}
// The arrays of a key built by groth16_setup in the layout groth16_pk_import takes (tests: export -> import round trip):
// which = 0..4 A, B, K, Z, B2; 5 / 6 idxA / idxB (uint32 packed); 7 alpha1 | beta1 | delta1 | beta2 | delta2.
template <class SU> static int32_t groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words) {
    typedef typename SU::G1T G1;
    typedef typename SU::G2T G2;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    constexpr size_t A1 = G1::AW / 2, A2 = G2::AW / 2;
    TRY(enter(ctx));
    if (!pk || !n_words) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pk->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "proving key belongs to another curve");
    const HostKey<SU>& hk = *(const HostKey<SU>*)pk->hk;
    const dvn_points* sets[5] = {pk->A, pk->B, pk->K, pk->Z, pk->B2};
    if (which >= 0 && which < 5) {
        const size_t per = which == 4 ? A2 : A1, cnt = sets[which]->n;
        *n_words = cnt * per;
        if (!out) return DVN_OK;
        if (cap_words < cnt * per) return fail(DVN_ERR_INVALID_ARG, "buffer too small");
        std::lock_guard<std::mutex> lk(ctx->mu);
        return which == 4 ? points_download<G2>(ctx, sets[4], 0, cnt, out) : points_download<G1>(ctx, sets[which], 0, cnt, out);
    }
    if (which == 5 || which == 6) {
        const size_t cnt = which == 5 ? pk->nA : pk->nB;
        *n_words = (cnt + 1) / 2;
        if (!out) return DVN_OK;
        if (cap_words < (cnt + 1) / 2) return fail(DVN_ERR_INVALID_ARG, "buffer too small");
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIP_TRY(hipMemcpy(out, which == 5 ? pk->d_idxA : pk->d_idxB, cnt * 4, hipMemcpyDeviceToHost));
        return DVN_OK;
    }
    if (which == 7) {
        *n_words = 3 * A1 + 2 * A2;
        if (!out) return DVN_OK;
        if (cap_words < 3 * A1 + 2 * A2) return fail(DVN_ERR_INVALID_ARG, "buffer too small");
        uint64_t j1[3 * A1 / 2 + 1], j2[3 * A2 / 2 + 1];
        H1::write_jac(hk.alpha, j1); memcpy(out, j1, A1 * 8); H1::write_jac(hk.beta, j1); memcpy(out + A1, j1, A1 * 8); H1::write_jac(hk.delta, j1); memcpy(out + 2 * A1, j1, A1 * 8);
        H2::write_jac(hk.beta2, j2); memcpy(out + 3 * A1, j2, A2 * 8); H2::write_jac(hk.delta2, j2); memcpy(out + 3 * A1 + A2, j2, A2 * 8);
        return DVN_OK;
    }
    return fail(DVN_ERR_INVALID_ARG, "which must be 0..7");
}
// Prove from the solver's output (gnark: `solution.W`, `solution.A/B/C`): w = m wire values, a / b / c = len <= n values each.
// Everything after the R1CS solve of groth16.Prove runs here: computeH, the filtered MSMs on four lanes, blinding, assembly.
template <class SU> static int32_t groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                                        const uint64_t* c, size_t len, const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) {
    typedef typename SU::FrC S;
    constexpr size_t EB = (size_t)S::EW * 4;
    TRY(enter(ctx));
    if (!pk || !w || !a || !b || !c || !r_mont || !s_mont || !proof_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pk->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "proving key belongs to another curve");
    if (len > pk->n) return fail(DVN_ERR_INVALID_ARG, "len = %zu exceeds the domain size %zu", len, pk->n);
    // a Setup-built key with a commitment knows its committed wires (K skips them by position) but this entry point takes the
    // solver's finished (w, a, b, c): only imported keys, whose K index list already leaves the committed wires out, qualify
    if (pk->n_committed && !pk->imported) return fail(DVN_ERR_INVALID_ARG, "the key carries a BSB22 commitment: use dvn_groth16_prove_committed");
    const HostKey<SU>& hk = *(const HostKey<SU>*)pk->hk;
    std::lock_guard<std::mutex> lk(ctx->mu);
    Blind<SU> bl(hk, r_mont, s_mont);
    // w goes up first (the four wire-only MSMs start on it), a / b / c follow while those run; every copy has completed when
    // groth16_device returns (it synchronises), so the caller's buffers are free on return (cgo rule)
    const uint64_t* src[3] = {a, b, c};
    MsmResults<SU> R;
    TRY(groth16_device<SU>(ctx, pk, w, 0u, 0, 1, &R, src, len));
    groth16_assemble<SU>(hk, R, bl, proof_out);
    return DVN_OK;
}

// ---- sharded proving: rank `rank` of `world` computes its rows of the five MSMs; the partial results travel as Jacobian
// images (affine normal form, Z = 1 or 0): A | B1 | K | Z (3 x fp words each) | B2 (3 x G2 coordinate words) ----------------
template <class SU> struct PartialLayout {
    static constexpr size_t J1 = SU::G1T::AW / 2 * 3 / 2, J2 = SU::G2T::AW / 2 * 3 / 2;     // uint64 words of a G1 / G2 Jacobian image
    static constexpr size_t WORDS = 4 * J1 + J2;
};
template <class SU> static int32_t groth16_prove_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                                       uint32_t rank, uint32_t world, uint64_t* partial_out) {
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    TRY(enter(ctx));
    if (!pk || !witness || !partial_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pk->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "proving key belongs to another curve");
    if (world < 1 || world > 64 || rank >= world) return fail(DVN_ERR_INVALID_ARG, "rank %u outside world %u", rank, world);
    if (pk->imported) return fail(DVN_ERR_INVALID_ARG, "imported key: it has no constraint system");
    if (pk->n_committed) return fail(DVN_ERR_INVALID_ARG, "the key carries a BSB22 commitment: the plain prover would leave the commitment wire unset");
    std::lock_guard<std::mutex> lk(ctx->mu);
    MsmResults<SU> R;
    TRY(groth16_device<SU>(ctx, pk, witness, flags & 1u, rank, world, &R));
    constexpr size_t J1 = PartialLayout<SU>::J1;
    H1::write_jac(R.mA, partial_out); H1::write_jac(R.mB, partial_out + J1); H1::write_jac(R.mK, partial_out + 2 * J1);
    H1::write_jac(R.mZ, partial_out + 3 * J1); H2::write_jac(R.mB2, partial_out + 4 * J1);
    return DVN_OK;
}
// the same share from the solver's output (gnark's own keys, imported or loaded: they have no constraint system here)
template <class SU> static int32_t groth16_prove_solved_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                                              const uint64_t* c, size_t len, uint32_t rank, uint32_t world, uint64_t* partial_out) {
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    TRY(enter(ctx));
    if (!pk || !w || !a || !b || !c || !partial_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (pk->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "proving key belongs to another curve");
    if (world < 1 || world > 64 || rank >= world) return fail(DVN_ERR_INVALID_ARG, "rank %u outside world %u", rank, world);
    if (len > pk->n) return fail(DVN_ERR_INVALID_ARG, "len = %zu exceeds the domain size %zu", len, pk->n);
    if (pk->n_committed && !pk->imported) return fail(DVN_ERR_INVALID_ARG, "the key carries a BSB22 commitment: use dvn_groth16_prove_committed_shard");
    std::lock_guard<std::mutex> lk(ctx->mu);
    const uint64_t* src[3] = {a, b, c};
    MsmResults<SU> R;
    TRY(groth16_device<SU>(ctx, pk, w, 0u, rank, world, &R, src, len));
    constexpr size_t J1 = PartialLayout<SU>::J1;
    H1::write_jac(R.mA, partial_out); H1::write_jac(R.mB, partial_out + J1); H1::write_jac(R.mK, partial_out + 2 * J1);
    H1::write_jac(R.mZ, partial_out + 3 * J1); H2::write_jac(R.mB2, partial_out + 4 * J1);
    return DVN_OK;
}
// host only: partials = world x PartialLayout::WORDS as gathered from the ranks
template <class SU> static int32_t groth16_prove_combine(const dvn_groth16_pk* pk, const uint64_t* partials, uint32_t world,
                                                         const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) {
    typedef typename SU::G1T G1;
    typedef typename SU::G2T G2;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    if (!pk || !partials || !r_mont || !s_mont || !proof_out || world < 1) return fail(DVN_ERR_INVALID_ARG, "bad argument");
    if (pk->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "proving key belongs to another curve");
    const HostKey<SU>& hk = *(const HostKey<SU>*)pk->hk;
    Blind<SU> bl(hk, r_mont, s_mont);
    auto rd1 = [](const uint64_t* j) -> typename H1::X {
        const u32* w = (const u32*)j; bool z = true;
        for (int i = 0; i < G1::AW / 2; ++i) z = z && w[G1::AW + i] == 0;                  // Z == 0: infinity
        if (z) return H1::infinity();
        u32 img[G1::AW]; G1::store_affine_internal(G1::affine_from_ext(w), img);
        return H1::load_device_affine(img);
    };
    auto rd2 = [](const uint64_t* j) -> typename H2::X {
        const u32* w = (const u32*)j; bool z = true;
        for (int i = 0; i < G2::AW / 2; ++i) z = z && w[G2::AW + i] == 0;
        if (z) return H2::infinity();
        u32 img[G2::AW]; G2::store_affine_internal(G2::affine_from_ext(w), img);
        return H2::load_device_affine(img);
    };
    constexpr size_t J1 = PartialLayout<SU>::J1, PW = PartialLayout<SU>::WORDS;
    MsmResults<SU> R{H1::infinity(), H1::infinity(), H1::infinity(), H1::infinity(), H2::infinity()};
    for (uint32_t k = 0; k < world; ++k) {
        const uint64_t* p = partials + (size_t)k * PW;
        R.mA = H1::add(R.mA, rd1(p)); R.mB = H1::add(R.mB, rd1(p + J1)); R.mK = H1::add(R.mK, rd1(p + 2 * J1));
        R.mZ = H1::add(R.mZ, rd1(p + 3 * J1)); R.mB2 = H2::add(R.mB2, rd2(p + 4 * J1));
    }
    groth16_assemble<SU>(hk, R, bl, proof_out);
    return DVN_OK;
}

// ---- one proof spread over the devices of a multi-device context (dvn_init(devs, ndev > 1)) -------------------------------------
// keys[i] is the SAME proving key resident on device i (dvn_ctx_device(ctx, i)): built by Setup or imported there.  One host
// thread per device runs that device's share of the five MSMs (groth16_device with rank i of ndev: replicated wire filtering /
// computeH, the device's entry range of every expanded key set); the partial results come back through pinned host memory and
// are added on the host (five group additions per device), then blinded and assembled as usual.  No collective is involved:
// one process sees all partial results, and the whole exchange is 5 points per device.  The statetransition proof is the
// node's serialisation point (/root/reference/sequencer/statetransition.go:80): this is the call that shortens it.
// prepare (optional): run on every device, with that device's context locked, before its share -- the solver-side steps of a
// committed proof; when given, the share reads the completed witness the hook left in the key's d_w.
struct NoPrepare { int32_t operator()(dvn_ctx*, dvn_groth16_pk*, uint32_t) const { return DVN_OK; } static constexpr bool active = false; };
template <class T> struct PrepareActive { static constexpr bool value = true; };
template <> struct PrepareActive<NoPrepare> { static constexpr bool value = false; };
template <class SU, class Prepare = NoPrepare>
static int32_t groth16_prove_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* witness, uint32_t flags,
                                                       const uint64_t* const* abc_host, size_t abc_len,
                                                       const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out, Prepare prepare = Prepare()) {
    constexpr bool prepared = PrepareActive<Prepare>::value;
    typedef typename SU::H1 H1;
    typedef typename SU::H2 H2;
    if (!ctx) return fail(DVN_ERR_INVALID_ARG, "ctx is NULL");
    if (!keys || !witness || !r_mont || !s_mont || !proof_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const uint32_t ndev = 1 + (uint32_t)ctx->peers.size();
    if (nkeys != ndev) return fail(DVN_ERR_INVALID_ARG, "%u keys for a context of %u devices: one key per device", nkeys, ndev);
    if (flags & 1u) return fail(DVN_ERR_INVALID_ARG, "multi-device proving takes the witness from host memory (every device uploads its own copy)");
    for (uint32_t i = 0; i < ndev; ++i) {
        dvn_groth16_pk* pk = keys[i];
        dvn_ctx* ci = i == 0 ? ctx : ctx->peers[i - 1];
        if (!pk) return fail(DVN_ERR_INVALID_ARG, "keys[%u] is NULL", i);
        if (pk->suite != SU::SUITE) return fail(DVN_ERR_INVALID_ARG, "keys[%u] belongs to another curve", i);
        if (pk->device != ci->device) return fail(DVN_ERR_INVALID_ARG, "keys[%u] lives on device %d, the context's device %u is %d", i, pk->device, i, ci->device);
        if (pk->m != keys[0]->m || pk->n != keys[0]->n) return fail(DVN_ERR_INVALID_ARG, "keys[%u] is not the same key as keys[0]", i);
        if (!abc_host && pk->imported) return fail(DVN_ERR_INVALID_ARG, "imported key: prove from the solver's output (prove_solved_multi)");
        if (pk->n_committed && !pk->imported && !prepared) return fail(DVN_ERR_INVALID_ARG, "the key carries a BSB22 commitment: use dvn_groth16_prove_committed_multi");
        if (abc_host && abc_len > pk->n) return fail(DVN_ERR_INVALID_ARG, "len = %zu exceeds the domain size %zu", abc_len, pk->n);
    }
    const HostKey<SU>& hk = *(const HostKey<SU>*)keys[0]->hk;
    Blind<SU> bl(hk, r_mont, s_mont);
    std::vector<MsmResults<SU>> parts(ndev);
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
                RoctxRange rr("dvn:device_share");
                if constexpr (prepared) {
                    rc = prepare(ci, keys[i], i);
                    if (rc == DVN_OK) rc = groth16_device<SU>(ci, keys[i], (const uint64_t*)keys[i]->d_w, 1u | 2u, i, ndev, &parts[i]);
                } else rc = groth16_device<SU>(ci, keys[i], witness, 0u, i, ndev, &parts[i], abc_host, abc_len);
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
    RoctxRange rr("dvn:host_combine");
    MsmResults<SU> R{H1::infinity(), H1::infinity(), H1::infinity(), H1::infinity(), H2::infinity()};
    for (uint32_t i = 0; i < ndev; ++i) {
        R.mA = H1::add(R.mA, parts[i].mA); R.mB = H1::add(R.mB, parts[i].mB); R.mK = H1::add(R.mK, parts[i].mK);
        R.mZ = H1::add(R.mZ, parts[i].mZ); R.mB2 = H2::add(R.mB2, parts[i].mB2);
    }
    groth16_assemble<SU>(hk, R, bl, proof_out);
    return DVN_OK;
}

// ---- per-proof statistics as JSON (what the reference logs with log.DebugTime around a proof,
// /root/reference/circuits/artifacts.go:572-577, plus what a device prover can add: stage times, bytes moved, achieved rate) ---
static inline int32_t groth16_stats_json(const dvn_groth16_pk* pk, char* buf, size_t cap, size_t* needed) {
    const size_t sb = pk->scalar_bytes, a1 = pk->g1_bytes, a2 = pk->g2_bytes;
    static const char* names[5] = {"G1.A", "G1.B", "G1.K", "G1.Z", "G2.B"};
    // algorithmic bytes: one scalar + one affine point per MSM pair, 7 transforms of (read + write) and the 4-vector pointwise
    // pass of computeH
    double alg = 0;
    for (int i = 0; i < 5; ++i) alg += (double)pk->msm_pairs[i] * (double)(sb + (i == 4 ? a2 : a1));
    alg += 7.0 * 2.0 * (double)sb * (double)pk->n + 4.0 * (double)sb * (double)pk->n;
    std::string o = "{";
    char t[256];
    snprintf(t, sizeof(t), "\"suite\": %d, \"log_n\": %u, \"wires\": %zu, \"rank\": %u, \"world\": %u, ", pk->suite, pk->log_n, pk->m, pk->last_rank, pk->last_world); o += t;
    snprintf(t, sizeof(t), "\"device_ms\": {\"until_h_ready\": %.4f, \"after_h\": %.4f, \"total\": %.4f}, ", pk->ms_h, pk->ms_msm, pk->ms_total); o += t;
    o += "\"msm\": {";
    for (int i = 0; i < 5; ++i) { snprintf(t, sizeof(t), "%s\"%s\": {\"pairs\": %zu, \"device_ms\": %.4f}", i ? ", " : "", names[i], pk->msm_pairs[i], pk->msm_ms[i]); o += t; }
    snprintf(t, sizeof(t), "}, \"bytes\": {\"host_to_device\": %zu, \"algorithmic\": %.0f}, \"achieved_GBps\": %.3f}", pk->h2d_bytes, alg,
             pk->ms_total > 0 ? alg / (pk->ms_total * 1e-3) / 1e9 : 0.0); o += t;
    if (needed) *needed = o.size() + 1;
    if (!buf) return DVN_OK;
    if (cap < o.size() + 1) return fail(DVN_ERR_INVALID_ARG, "buffer holds %zu bytes, the statistics need %zu", cap, o.size() + 1);
    memcpy(buf, o.c_str(), o.size() + 1);
    return DVN_OK;
}

// ---- Pedersen commitment + proof of knowledge on any curve (gnark-crypto ecc/<curve>/fr/pedersen ProvingKey.Commit /
// ProveKnowledge behind the BSB22 commitment hint of groth16.Prove): two G1 MSMs over the committed wire values, in flight
// together.  values: n scalar images (host, or device with flags & 1).  Outputs: G1Affine images of the curve.
template <class SU> static int32_t pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma,
                                                   const uint64_t* values_mont, size_t n, uint32_t flags, uint64_t* commitment_out, uint64_t* pok_out) {
    typedef typename SU::FrC S;
    typedef typename SU::H1 H1;
    constexpr size_t SB = (size_t)S::EW * 4, A1 = SU::G1T::AW / 2;
    TRY(enter(ctx));
    if (!basis || !basis_exp_sigma || (!values_mont && n) || !commitment_out || !pok_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (basis->curve != 1 || basis_exp_sigma->curve != 1 || basis->suite != SU::SUITE || basis_exp_sigma->suite != SU::SUITE)
        return fail(DVN_ERR_INVALID_ARG, "bases must be G1 point sets of this curve");
    if (n > basis->n || n > basis_exp_sigma->n) return fail(DVN_ERR_INVALID_ARG, "len(values) exceeds the basis");
    std::lock_guard<std::mutex> lk(ctx->mu);
    u32* d_vals = nullptr;
    const u32* vals = (const u32*)values_mont;
    if (!(flags & 1) && n) {               // host values: one upload shared by both MSMs
        HIP_TRY(hipMalloc((void**)&d_vals, n * SB));
        hipError_t e = hipMemcpy(d_vals, values_mont, n * SB, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d_vals); (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "upload failed: %s", hipGetErrorString(e)); }
        vals = d_vals;
    }
    MsmLane *la = nullptr, *lb = nullptr;
    typename H1::X c, k;
    int32_t rc = SU::msm_submit_g1(ctx, basis, 0, n, vals, &la);
    if (rc == DVN_OK) rc = SU::msm_submit_g1(ctx, basis_exp_sigma, 0, n, vals, &lb);
    if (rc == DVN_OK) rc = SU::msm_collect_g1(ctx, la, &c); else if (la) { typename H1::X t; (void)SU::msm_collect_g1(ctx, la, &t); }
    if (rc == DVN_OK) rc = SU::msm_collect_g1(ctx, lb, &k); else if (lb && lb->busy) { typename H1::X t; (void)SU::msm_collect_g1(ctx, lb, &t); }
    if (d_vals) (void)hipFree(d_vals);
    if (rc != DVN_OK) return rc;
    uint64_t j[3 * A1 / 2 + 1];
    H1::write_jac(c, j); memcpy(commitment_out, j, A1 * 8); if (H1::is_inf(c)) memset(commitment_out, 0, A1 * 8);
    H1::write_jac(k, j); memcpy(pok_out, j, A1 * 8); if (H1::is_inf(k)) memset(pok_out, 0, A1 * 8);
    return DVN_OK;
}
