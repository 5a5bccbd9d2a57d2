// Groth16 (BN254) host orchestration over the device primitives: the C++ restatement of what the Go shim's
// hipGroth16Prove does, runnable without a Go toolchain.
//
// Mirrors, step for step, gnark backend/groth16/bn254 prove.go `Prove` (called from
// /root/reference/prover/prover_cpu.go:37 and, through the ICICLE wrapper, prover/prover_gpu.go:33) and
// setup.go `Setup` (row a9; /root/reference/prover/setup.go:15-28) for a plain sparse R1CS:
//   Setup:  toxic waste -> Lagrange basis at tau -> A_i(tau), B_i(tau), C_i(tau) -> K, Z (bit-reversed) ->
//           fixed-base batch scalar multiplications -> proving key arrays resident in HBM, infinity filtering
//   Prove:  (a, b, c) = (A w, B w, C w)  ->  h = computeH  ->  wire filtering by InfinityA/B  ->
//           Ar, Bs1 (G1 MSM), Bs (G2 MSM), Krs (2 G1 MSMs)  ->  blinding with (r, s)  ->  affine proof
// The R1CS solver proper (hints, emulated arithmetic) stays in gnark on the CPU; here the "solve" is the sparse
// matrix-vector product over an already complete wire assignment.
#include <algorithm>
#include "dvn_internal.h"
#include "msm_kernels.h"
#include "ntt_kernels.h"

// ------------------------------------------------------------------------------------------------------
// types
// ------------------------------------------------------------------------------------------------------
struct Csr {
    std::vector<u32> ptr, idx;      // rows -> (wire, coeff)
    std::vector<u32> coef;          // 8 words per entry, internal Montgomery form (packed)
};
struct dvn_r1cs {
    u32 n_wires = 0, n_public = 0, n_constraints = 0;
    Csr m[3];                       // A, B, C by constraint row
    std::vector<u32> witness;       // n_wires x 8, gnark Montgomery words (synthetic instances only)
};
struct dvn_groth16_pk {
    u32 log_n = 0;
    size_t n = 0, m = 0, n_public = 0, n_constraints = 0;
    dvn_domain* dom = nullptr;
    dvn_points *A = nullptr, *B = nullptr, *K = nullptr, *Z = nullptr, *B2 = nullptr;
    size_t nA = 0, nB = 0;
    u32 *d_idxA = nullptr, *d_idxB = nullptr;           // wires kept after infinity filtering (InfinityA/B)
    dvn_host::HG1::X alpha, beta, delta;       // key constants in the host finishing arithmetic (host_fast.h)
    dvn_host::HG2::X beta2, delta2;
    u32 *d_ptr[3] = {nullptr, nullptr, nullptr}, *d_idx[3] = {nullptr, nullptr, nullptr}, *d_coef[3] = {nullptr, nullptr, nullptr};
    u32 *d_w = nullptr, *d_abc = nullptr, *d_wA = nullptr, *d_wB = nullptr;   // per-proof buffers, allocated once
    float ms_h = 0, ms_msm = 0, ms_total = 0;
};

// ------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------

// out[row] = sum_k coef[k] * w[idx[k]]; w are gnark Montgomery words (read as x*2^-5 internal), coef true internal:
// the product is again a gnark Montgomery word.
static __global__ void k_r1cs_matvec(const u32* __restrict__ ptr, const u32* __restrict__ idx, const u32* __restrict__ coef,
                                     const u32* __restrict__ w, u32* __restrict__ out, u32 rows) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    Fr acc = fe_zero<FrCfg>();
    for (u32 k = ptr[r]; k < ptr[r + 1]; ++k)
        acc = fe_add<FrCfg>(acc, fe_mul<FrCfg>(ld_fr<FrCfg>(coef + (size_t)k * 8), ld_fr<FrCfg>(w + (size_t)idx[k] * 8)));
    st_fr<FrCfg>(acc, out + (size_t)r * 8);
}
static __global__ void k_gather_fr(const u32* __restrict__ src, const u32* __restrict__ idx, u32* __restrict__ dst, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)idx[i] * 8);
    uint4* d = reinterpret_cast<uint4*>(dst + i * 8);
    d[0] = s[0]; d[1] = s[1];
}
// L_j(tau) = (tau^n - 1)/n * w^j / (tau - w^j); consts = {w, tau, (tau^n - 1)/n}, all internal form
static __global__ void k_lagrange(u32* __restrict__ out, const u32* __restrict__ consts, size_t n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const Fr w = ld_fr<FrCfg>(consts), tau = ld_fr<FrCfg>(consts + 8), zn = ld_fr<FrCfg>(consts + 16);
    const Fr wj = fe_pow_u64<FrCfg>(w, (u64)j);
    const Fr den = fe_inv<FrCfg>(fe_sub<FrCfg>(tau, wj));
    st_fr<FrCfg>(fe_mul<FrCfg>(fe_mul<FrCfg>(zn, wj), den), out + j * 8);
}
// column evaluation: out[i] = sum_k coef[k] * L[row[k]] over the entries of wire i (CSC of one R1CS matrix)
static __global__ void k_col_eval(const u32* __restrict__ ptr, const u32* __restrict__ row, const u32* __restrict__ coef,
                                  const u32* __restrict__ L, u32* __restrict__ out, u32 cols) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cols) return;
    Fr acc = fe_zero<FrCfg>();
    for (u32 k = ptr[i]; k < ptr[i + 1]; ++k)
        acc = fe_add<FrCfg>(acc, fe_mul<FrCfg>(ld_fr<FrCfg>(coef + (size_t)k * 8), ld_fr<FrCfg>(L + (size_t)row[k] * 8)));
    st_fr<FrCfg>(acc, out + (size_t)i * 8);
}
// K_i = (beta * At_i + alpha * Bt_i + Ct_i) / delta for i >= n_public; consts = {alpha, beta, 1/delta}
static __global__ void k_key_k(const u32* __restrict__ at, const u32* __restrict__ bt, const u32* __restrict__ ct,
                               const u32* __restrict__ consts, u32* __restrict__ out, u32 first, u32 count) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t i = (size_t)first + t;
    const Fr alpha = ld_fr<FrCfg>(consts), beta = ld_fr<FrCfg>(consts + 8), dinv = ld_fr<FrCfg>(consts + 16);
    Fr v = fe_add<FrCfg>(fe_add<FrCfg>(fe_mul<FrCfg>(beta, ld_fr<FrCfg>(at + i * 8)), fe_mul<FrCfg>(alpha, ld_fr<FrCfg>(bt + i * 8))), ld_fr<FrCfg>(ct + i * 8));
    st_fr<FrCfg>(fe_mul<FrCfg>(v, dinv), out + (size_t)t * 8);
}
// Z'_i = zdt * tau^(bitrev(i)): gnark computes Z_i = (tau^n - 1)/delta * tau^i and bit-reverses the POINT array
static __global__ void k_key_z(u32* __restrict__ out, const u32* __restrict__ consts /* {tau, zdt} */, u32 log_n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const Fr tau = ld_fr<FrCfg>(consts), zdt = ld_fr<FrCfg>(consts + 8);
    const u32 e = __brev(i) >> (32 - log_n);
    st_fr<FrCfg>(fe_mul<FrCfg>(zdt, fe_pow_u64<FrCfg>(tau, (u64)e)), out + (size_t)i * 8);
}
// fixed-base table: table[(w * 256 + d)] = d * 2^(8w) * G (affine internal), d in 1..255, w in 0..31
template <class Cv> __global__ void k_fb_table(u32* __restrict__ table, const u32* __restrict__ gen) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 32 * 256) return;
    const u32 w = t >> 8, d = t & 255;
    typename Cv::X p = Cv::from_affine(ld_affine<Cv>(gen));
    for (u32 k = 0; k < 8 * w; ++k) p = Cv::dbl(p);
    typename Cv::A a = Cv::to_affine(Cv::mul_small(p, d));
    u32 wds[Cv::AW];
    Cv::store_affine_internal(a, wds);
    store_words16(table + (size_t)t * Cv::AW, wds, Cv::AW);
}
// out[i] = [k_i] G via the table (gnark's BatchScalarMultiplicationG1/G2 at Setup); scalars internal form
template <class Cv> __global__ void k_fixed_base(const u32* __restrict__ table, const u32* __restrict__ scalars,
                                                 u32* __restrict__ out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u32 kw[8];
    fe_pack<FrCfg>(fe_canonical<FrCfg>(fe_mul<FrCfg>(ld_fr<FrCfg>(scalars + i * 8), fe_small<FrCfg>(1))), kw);   // internal -> integer
    typename Cv::X acc = Cv::infinity();
    for (u32 w = 0; w < 32; ++w) {
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
    // canonical Fr as internal-form element (oracle/bn254_py.py SplitMix64.next_fr)
    Fr next_fr() {
        u32 w[8];
        for (int k = 0; k < 4; ++k) { u64 v = next(); w[2 * k] = (u32)v; w[2 * k + 1] = (u32)(v >> 32); }
        while (geq8(w, FR_MOD_W)) sub8(w, FR_MOD_W);
        return fe_from_canonical<FrCfg>(fe_unpack<FrCfg>(w));
    }
};
static Fr fr_u64(u64 v) { u32 w[8] = {(u32)v, (u32)(v >> 32), 0, 0, 0, 0, 0, 0}; return fe_from_canonical<FrCfg>(fe_unpack<FrCfg>(w)); }
static void fr_pack_internal(const Fr& v, u32* w) { u32 t[8]; fe_pack<FrCfg>(fe_canonical<FrCfg>(v), t); memcpy(w, t, 32); }
static void fr_to_ext_words(const Fr& v, u32* w) { u32 t[8]; fe_to_ext_mont<FrCfg>(v, t); memcpy(w, t, 32); }
static void fr_canonical_words(const Fr& v, u32* w) { u32 t[8]; fe_pack<FrCfg>(fe_canonical<FrCfg>(fe_mul<FrCfg>(v, fe_small<FrCfg>(1))), t); memcpy(w, t, 32); }
static Fr fr_from_ext_words(const u64* p) { u32 t[8]; memcpy(t, p, 32); return fe_from_ext_mont<FrCfg>(t); }
static bool fr_is_zero_h(const Fr& v) { return fe_is_zero<FrCfg>(v); }

template <class T> static int32_t dev_upload(T** d, const std::vector<T>& h) {
    HIP_TRY(hipMalloc((void**)d, h.size() * sizeof(T) + 16));
    if (!h.empty()) HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return DVN_OK;
}

// ------------------------------------------------------------------------------------------------------
// synthetic R1CS (oracle/bn254_py.py SyntheticR1CS): rows (w[l1] + k*w[l2]) * w[r] = w[o], witness built forward
// ------------------------------------------------------------------------------------------------------
extern "C" int32_t dvn_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out) {
    if (!out || n_public < 1) return fail(DVN_ERR_INVALID_ARG, "bad argument");
    dvn_r1cs* cs = new (std::nothrow) dvn_r1cs();
    if (!cs) return fail(DVN_ERR_OOM, "host allocation failed");
    SplitMix rng(seed);
    const u32 n_free = 4;
    std::vector<Fr> w;
    w.reserve((size_t)n_public + n_free + n_constraints);
    w.push_back(fe_one<FrCfg>());
    for (u32 i = 1; i < n_public; ++i) w.push_back(rng.next_fr());
    for (u32 i = 0; i < n_free; ++i) w.push_back(rng.next_fr());
    for (int k = 0; k < 3; ++k) cs->m[k].ptr.push_back(0);
    auto push = [&](Csr& c, u32 wire, u64 coefv) { c.idx.push_back(wire); u32 t[8]; fr_pack_internal(fr_u64(coefv), t); c.coef.insert(c.coef.end(), t, t + 8); };
    for (u32 j = 0; j < n_constraints; ++j) {
        const u64 mcur = w.size();
        const u32 l1 = (u32)(rng.next() % mcur), l2 = (u32)(rng.next() % mcur);
        const u64 k = rng.next() % 7 + 1;
        const u32 r_ = (u32)(rng.next() % mcur), o = (u32)mcur;
        Fr val = fe_mul<FrCfg>(fe_add<FrCfg>(w[l1], fe_mul<FrCfg>(fr_u64(k), w[l2])), w[r_]);
        w.push_back(val);
        if (l1 != l2) { push(cs->m[0], l1, 1); push(cs->m[0], l2, k); } else push(cs->m[0], l1, 1 + k);
        push(cs->m[1], r_, 1);
        push(cs->m[2], o, 1);
        for (int q = 0; q < 3; ++q) cs->m[q].ptr.push_back((u32)cs->m[q].idx.size());
    }
    cs->n_wires = (u32)w.size(); cs->n_public = n_public; cs->n_constraints = n_constraints;
    cs->witness.resize((size_t)cs->n_wires * 8);
    for (size_t i = 0; i < w.size(); ++i) fr_to_ext_words(w[i], cs->witness.data() + i * 8);
    *out = cs;
    return DVN_OK;
}
extern "C" int32_t dvn_r1cs_info(const dvn_r1cs* cs, uint32_t* n_wires, uint32_t* n_public, uint32_t* n_constraints) {
    if (!cs) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (n_wires) *n_wires = cs->n_wires;
    if (n_public) *n_public = cs->n_public;
    if (n_constraints) *n_constraints = cs->n_constraints;
    return DVN_OK;
}
extern "C" int32_t dvn_r1cs_witness(const dvn_r1cs* cs, uint64_t* out) {
    if (!cs || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    memcpy(out, cs->witness.data(), cs->witness.size() * 4);
    return DVN_OK;
}
extern "C" void dvn_r1cs_free(dvn_r1cs* cs) { delete cs; }

// ------------------------------------------------------------------------------------------------------
// Setup
// ------------------------------------------------------------------------------------------------------
extern "C" int32_t dvn_groth16_pk_free(dvn_ctx* ctx, dvn_groth16_pk* pk) {
    TRY(enter(ctx));
    if (!pk) return DVN_OK;
    dvn_points* ps[] = {pk->A, pk->B, pk->K, pk->Z, pk->B2};
    for (dvn_points* p : ps) if (p) (void)dvn_points_free(ctx, p);
    if (pk->dom) (void)dvn_ntt_domain_free(ctx, pk->dom);
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipStreamSynchronize(ctx->stream);
    u32* ptrs[] = {pk->d_idxA, pk->d_idxB, pk->d_ptr[0], pk->d_ptr[1], pk->d_ptr[2], pk->d_idx[0], pk->d_idx[1], pk->d_idx[2],
                   pk->d_coef[0], pk->d_coef[1], pk->d_coef[2], pk->d_w, pk->d_abc, pk->d_wA, pk->d_wB};
    for (u32* p : ptrs) if (p) (void)hipFree(p);
    delete pk;
    return DVN_OK;
}

template <class Cv>
static int32_t fixed_base_points(dvn_ctx* ctx, const typename Cv::A& gen, const u32* d_scalars, size_t count, int curve, dvn_points** out) {
    dvn_points* p = new (std::nothrow) dvn_points();
    if (!p) return fail(DVN_ERR_OOM, "host allocation failed");
    p->curve = curve; p->n = count;
    expand_plan(count, 254, (size_t)Cv::AW * 4, &p->expand_c, &p->nexp);
    u32 *table = nullptr, *dgen = nullptr;
    u32 hg[Cv::AW];
    Cv::store_affine_internal(gen, hg);
    hipError_t e = hipMalloc((void**)&p->d, count ? count * Cv::AW * 4 * p->nexp : 16);
    if (e == hipSuccess) e = hipMalloc((void**)&table, (size_t)32 * 256 * Cv::AW * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dgen, sizeof(hg));
    if (e == hipSuccess) e = hipMemcpy(dgen, hg, sizeof(hg), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_fb_table<Cv><<<32 * 256 / 64, 64, 0, ctx->stream>>>(table, dgen);
        if (count) k_fixed_base<Cv><<<(unsigned)((count + 63) / 64), 64, 0, ctx->stream>>>(table, d_scalars, p->d, count);
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
extern "C" int32_t dvn_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out) {
    TRY(enter(ctx));
    if (!cs || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (cs->n_constraints < 2) return fail(DVN_ERR_INVALID_ARG, "need at least 2 constraints");
    u32 log_n = 1;
    while (((size_t)1 << log_n) < cs->n_constraints) ++log_n;
    const size_t n = (size_t)1 << log_n, m = cs->n_wires;
    dvn_groth16_pk* pk = new (std::nothrow) dvn_groth16_pk();
    if (!pk) return fail(DVN_ERR_OOM, "host allocation failed");
    pk->log_n = log_n; pk->n = n; pk->m = m; pk->n_public = cs->n_public; pk->n_constraints = cs->n_constraints;
    int32_t rc = dvn_ntt_domain(ctx, log_n, &pk->dom);
    if (rc) { delete pk; return rc; }
    std::unique_lock<std::mutex> lk(ctx->mu);
    SplitMix rng(seed);
    Fr tw[5];
    for (int i = 0; i < 5; ++i) { tw[i] = rng.next_fr(); if (fr_is_zero_h(tw[i])) tw[i] = fe_one<FrCfg>(); }
    const Fr tau = tw[0], alpha = tw[1], beta = tw[2], delta = tw[4];
    const Fr one = fe_one<FrCfg>(), dinv = fe_inv<FrCfg>(delta);
    const Fr ztau = fe_sub<FrCfg>(fe_pow_u64<FrCfg>(tau, n), one);
    // root of unity of order n, as in dvn_ntt.hip
    static const u64 root28[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL};
    Fr omega = fe_from_canonical<FrCfg>(fe_unpack<FrCfg>(*(const u32(*)[8])root28));
    for (u32 i = log_n; i < 28; ++i) omega = fe_sqr<FrCfg>(omega);

    u32 *d_L = nullptr, *d_t[3] = {nullptr, nullptr, nullptr}, *d_k = nullptr, *d_z = nullptr, *d_consts = nullptr;
    u32 *d_cptr = nullptr, *d_crow = nullptr, *d_ccoef = nullptr, *d_cmp = nullptr;
    auto cleanup = [&]() {
        u32* ps[] = {d_L, d_t[0], d_t[1], d_t[2], d_k, d_z, d_consts, d_cptr, d_crow, d_ccoef, d_cmp};
        for (u32* p : ps) if (p) (void)hipFree(p);
    };
    auto body = [&]() -> int32_t {
        hipStream_t st = ctx->stream;
        HIP_TRY(hipMalloc((void**)&d_L, n * 32));
        HIP_TRY(hipMalloc((void**)&d_consts, 8 * 32));
        for (int q = 0; q < 3; ++q) HIP_TRY(hipMalloc((void**)&d_t[q], m * 32));
        u32 hc[8 * 8];
        fr_pack_internal(omega, hc); fr_pack_internal(tau, hc + 8);
        fr_pack_internal(fe_mul<FrCfg>(ztau, fe_inv<FrCfg>(fr_u64(n))), hc + 16);
        fr_pack_internal(alpha, hc + 24); fr_pack_internal(beta, hc + 32); fr_pack_internal(dinv, hc + 40);
        fr_pack_internal(tau, hc + 48); fr_pack_internal(fe_mul<FrCfg>(ztau, dinv), hc + 56);
        HIP_TRY(hipMemcpy(d_consts, hc, sizeof(hc), hipMemcpyHostToDevice));
        k_lagrange<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(d_L, d_consts, n);
        HIP_TRY(hipGetLastError());
        // column (CSC) form of each matrix, evaluated at tau
        for (int q = 0; q < 3; ++q) {
            const Csr& r = cs->m[q];
            std::vector<u32> cptr(m + 1, 0), crow(r.idx.size()), ccoef(r.idx.size() * 8);
            for (u32 wi : r.idx) cptr[wi + 1]++;
            for (size_t i = 0; i < m; ++i) cptr[i + 1] += cptr[i];
            std::vector<u32> cur(cptr.begin(), cptr.end() - 1);
            for (u32 row = 0; row < cs->n_constraints; ++row)
                for (u32 k = r.ptr[row]; k < r.ptr[row + 1]; ++k) {
                    const u32 pos = cur[r.idx[k]]++;
                    crow[pos] = row; memcpy(&ccoef[(size_t)pos * 8], &r.coef[(size_t)k * 8], 32);
                }
            TRY(dev_upload(&d_cptr, cptr)); TRY(dev_upload(&d_crow, crow)); TRY(dev_upload(&d_ccoef, ccoef));
            k_col_eval<<<(unsigned)((m + 63) / 64), 64, 0, st>>>(d_cptr, d_crow, d_ccoef, d_L, d_t[q], (u32)m);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(d_cptr); (void)hipFree(d_crow); (void)hipFree(d_ccoef); d_cptr = d_crow = d_ccoef = nullptr;
            // row-major copy for Prove's matvec
            TRY(dev_upload(&pk->d_ptr[q], r.ptr)); TRY(dev_upload(&pk->d_idx[q], r.idx)); TRY(dev_upload(&pk->d_coef[q], r.coef));
        }
        // K (private wires) and Z (bit-reversed)
        const size_t nk = m - cs->n_public;
        HIP_TRY(hipMalloc((void**)&d_k, (nk ? nk : 1) * 32));
        HIP_TRY(hipMalloc((void**)&d_z, n * 32));
        if (nk) k_key_k<<<(unsigned)((nk + 63) / 64), 64, 0, st>>>(d_t[0], d_t[1], d_t[2], d_consts + 24, d_k, cs->n_public, (u32)nk);
        k_key_z<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(d_z, d_consts + 48, log_n);
        HIP_TRY(hipGetLastError());
        // infinity filtering of A and B (gnark's InfinityA / InfinityB)
        std::vector<u32> hA(m * 8), hB(m * 8);
        HIP_TRY(hipMemcpyAsync(hA.data(), d_t[0], m * 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hB.data(), d_t[1], m * 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<u32> idxA, idxB;
        auto nonzero = [](const u32* p) { u32 z = 0; for (int k = 0; k < 8; ++k) z |= p[k]; return z != 0; };
        for (u32 i = 0; i < m; ++i) { if (nonzero(&hA[(size_t)i * 8])) idxA.push_back(i); if (nonzero(&hB[(size_t)i * 8])) idxB.push_back(i); }
        pk->nA = idxA.size(); pk->nB = idxB.size();
        TRY(dev_upload(&pk->d_idxA, idxA)); TRY(dev_upload(&pk->d_idxB, idxB));
        const size_t cmax = std::max(pk->nA, pk->nB);
        HIP_TRY(hipMalloc((void**)&d_cmp, (cmax ? cmax : 1) * 32));
        const G1::A g1 = g1_generator();
        const G2::A g2 = g2_generator();
        if (pk->nA) k_gather_fr<<<(unsigned)((pk->nA + 255) / 256), 256, 0, st>>>(d_t[0], pk->d_idxA, d_cmp, pk->nA);
        HIP_TRY(hipGetLastError());
        TRY(fixed_base_points<G1>(ctx, g1, d_cmp, pk->nA, 1, &pk->A));
        if (pk->nB) k_gather_fr<<<(unsigned)((pk->nB + 255) / 256), 256, 0, st>>>(d_t[1], pk->d_idxB, d_cmp, pk->nB);
        HIP_TRY(hipGetLastError());
        TRY(fixed_base_points<G1>(ctx, g1, d_cmp, pk->nB, 1, &pk->B));
        TRY(fixed_base_points<G2>(ctx, g2, d_cmp, pk->nB, 2, &pk->B2));
        TRY(fixed_base_points<G1>(ctx, g1, d_k, nk, 1, &pk->K));
        TRY(fixed_base_points<G1>(ctx, g1, d_z, n, 1, &pk->Z));
        // per-proof buffers
        HIP_TRY(hipMalloc((void**)&pk->d_w, m * 32));
        HIP_TRY(hipMalloc((void**)&pk->d_abc, 3 * n * 32));
        HIP_TRY(hipMalloc((void**)&pk->d_wA, (pk->nA ? pk->nA : 1) * 32));
        HIP_TRY(hipMalloc((void**)&pk->d_wB, (pk->nB ? pk->nB : 1) * 32));
        // G1/G2 constants of the key
        u32 kw[8], img1[G1::AW], img2[G2::AW];
        G1::store_affine_internal(g1, img1); G2::store_affine_internal(g2, img2);
        const dvn_host::HG1::X hg1 = dvn_host::HG1::load_device_affine(img1);
        const dvn_host::HG2::X hg2 = dvn_host::HG2::load_device_affine(img2);
        fr_canonical_words(alpha, kw); pk->alpha = dvn_host::HG1::normalize(dvn_host::HG1::mul_scalar(hg1, kw));
        fr_canonical_words(beta, kw); pk->beta = dvn_host::HG1::normalize(dvn_host::HG1::mul_scalar(hg1, kw));
        pk->beta2 = dvn_host::HG2::normalize(dvn_host::HG2::mul_scalar(hg2, kw));
        fr_canonical_words(delta, kw); pk->delta = dvn_host::HG1::normalize(dvn_host::HG1::mul_scalar(hg1, kw));
        pk->delta2 = dvn_host::HG2::normalize(dvn_host::HG2::mul_scalar(hg2, kw));
        return DVN_OK;
    };
    rc = body();
    cleanup();
    lk.unlock();
    if (rc != DVN_OK) { std::string keep = tl_error; (void)dvn_groth16_pk_free(ctx, pk); tl_error = keep; return rc; }
    *out = pk;
    return DVN_OK;
}

extern "C" int32_t dvn_groth16_pk_info(const dvn_groth16_pk* pk, uint32_t* log_n, uint64_t* sizes /* nA, nB, nK, nZ, nB2 */) {
    if (!pk) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (log_n) *log_n = pk->log_n;
    if (sizes) { sizes[0] = pk->nA; sizes[1] = pk->nB; sizes[2] = pk->m - pk->n_public; sizes[3] = pk->n; sizes[4] = pk->nB; }
    return DVN_OK;
}

// ------------------------------------------------------------------------------------------------------
// Prove
// ------------------------------------------------------------------------------------------------------
// witness: m x 4 uint64 gnark Montgomery words (host, or device with flags & 1); r_mont/s_mont: the blinding
// factors gnark draws from crypto/rand, injected so results are reproducible.  proof_out: Ar (8) | Bs (16) | Krs (8)
// uint64, affine Montgomery images exactly as gnark's groth16_bn254.Proof holds them.
extern "C" int32_t dvn_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                     const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]) {
    TRY(enter(ctx));
    if (!pk || !witness || !r_mont || !s_mont || !proof_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t st = ctx->stream;
    const size_t n = pk->n, m = pk->m;
    hipEvent_t e0, e1, e2;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2));
    auto body = [&]() -> int32_t {
        HIP_TRY(hipEventRecord(e0, st));
        HIP_TRY(hipMemcpyAsync(pk->d_w, witness, m * 32, (flags & 1) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        // (1) "solve": a = A w, b = B w, c = C w, zero padded to the domain size
        HIP_TRY(hipMemsetAsync(pk->d_abc, 0, 3 * n * 32, st));
        for (int q = 0; q < 3; ++q)
            k_r1cs_matvec<<<(unsigned)((pk->n_constraints + 127) / 128), 128, 0, st>>>(pk->d_ptr[q], pk->d_idx[q], pk->d_coef[q], pk->d_w,
                                                                                      pk->d_abc + (size_t)q * n * 8, (u32)pk->n_constraints);
        HIP_TRY(hipGetLastError());
        // (3) h = computeH(a, b, c), left on the device as the scalar vector of the Z MSM
        TRY(dvn_internal_compute_h_dev(ctx, pk->dom, pk->d_abc, pk->d_abc + n * 8, pk->d_abc + 2 * n * 8));
        HIP_TRY(hipEventRecord(e1, st));
        // (4) filter the wire vector by InfinityA / InfinityB
        if (pk->nA) k_gather_fr<<<(unsigned)((pk->nA + 255) / 256), 256, 0, st>>>(pk->d_w, pk->d_idxA, pk->d_wA, pk->nA);
        if (pk->nB) k_gather_fr<<<(unsigned)((pk->nB + 255) / 256), 256, 0, st>>>(pk->d_w, pk->d_idxB, pk->d_wB, pk->nB);
        HIP_TRY(hipGetLastError());
        // (6) the five MSMs
        typedef dvn_host::HG1 H1;
        typedef dvn_host::HG2 H2;
        H1::X mA, mB, mK, mZ; H2::X mB2;
        // two MSMs in flight: the sort and the latency-bound tail of one run under the accumulate kernel of the other;
        // the G2 MSM (the longest) goes first and stays in flight while the G1 ones cycle through the second lane
        MsmLane *l2 = nullptr, *l1 = nullptr;
        TRY(dvn_internal_msm_submit_g2(ctx, pk->B2, 0, pk->nB, pk->d_wB, &l2));
        TRY(dvn_internal_msm_submit_g1(ctx, pk->A, 0, pk->nA, pk->d_wA, &l1));
        TRY(dvn_internal_msm_collect_g1(ctx, l1, &mA));
        TRY(dvn_internal_msm_submit_g1(ctx, pk->B, 0, pk->nB, pk->d_wB, &l1));
        TRY(dvn_internal_msm_collect_g1(ctx, l1, &mB));
        TRY(dvn_internal_msm_submit_g1(ctx, pk->K, 0, m - pk->n_public, pk->d_w + pk->n_public * 8, &l1));
        TRY(dvn_internal_msm_collect_g1(ctx, l1, &mK));
        TRY(dvn_internal_msm_submit_g1(ctx, pk->Z, 0, n - 1, pk->d_abc, &l1));               // h[:n-1] against Z[:n-1]
        TRY(dvn_internal_msm_collect_g1(ctx, l1, &mZ));
        TRY(dvn_internal_msm_collect_g2(ctx, l2, &mB2));
        HIP_TRY(hipEventRecord(e2, st));
        HIP_TRY(hipStreamSynchronize(st));
        // (5)(7) blinding and assembly on the host: O(1) group operations (host_fast.h)
        const Fr r = fr_from_ext_words(r_mont), s = fr_from_ext_words(s_mont);
        u32 rw[8], sw[8], krw[8];
        fr_canonical_words(r, rw); fr_canonical_words(s, sw);
        fr_canonical_words(fe_neg<FrCfg>(fe_mul<FrCfg>(r, s)), krw);
        const H1::X ar = H1::normalize(H1::add(H1::add(mA, pk->alpha), H1::mul_scalar(pk->delta, rw)));
        const H1::X bs1 = H1::normalize(H1::add(H1::add(mB, pk->beta), H1::mul_scalar(pk->delta, sw)));
        const H2::X bs = H2::add(H2::add(mB2, pk->beta2), H2::mul_scalar(pk->delta2, sw));
        H1::X krs = H1::add(mK, mZ);
        krs = H1::add(krs, H1::mul_scalar(pk->delta, krw));
        krs = H1::add(krs, H1::mul_scalar(ar, sw));
        krs = H1::add(krs, H1::mul_scalar(bs1, rw));
        uint64_t j1[12], j2[24];
        H1::write_jac(ar, j1); memcpy(proof_out, j1, 64);
        H2::write_jac(bs, j2); memcpy(proof_out + 8, j2, 128);
        H1::write_jac(krs, j1); memcpy(proof_out + 24, j1, 64);
        if (H1::is_inf(ar)) memset(proof_out, 0, 64);                 // gnark's affine infinity is (0, 0)
        if (H2::is_inf(bs)) memset(proof_out + 8, 0, 128);
        if (H1::is_inf(krs)) memset(proof_out + 24, 0, 64);
        HIP_TRY(hipEventElapsedTime(&pk->ms_h, e0, e1));
        HIP_TRY(hipEventElapsedTime(&pk->ms_msm, e1, e2));
        HIP_TRY(hipEventElapsedTime(&pk->ms_total, e0, e2));
        return DVN_OK;
    };
    int32_t rc = body();
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return rc;
}
extern "C" int32_t dvn_groth16_last_ms(const dvn_groth16_pk* pk, float* ms_h, float* ms_msm, float* ms_total) {
    if (!pk) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (ms_h) *ms_h = pk->ms_h;
    if (ms_msm) *ms_msm = pk->ms_msm;
    if (ms_total) *ms_total = pk->ms_total;
    return DVN_OK;
}

// ------------------------------------------------------------------------------------------------------
// Solidity ABI image of a proof: /root/reference/solidity/solidity.go:29-73,85-116
// (uint256[8] proof, uint256[2] commitments, uint256[2] commitmentPok), every word big-endian, G2 coordinates in
// EIP-197 order (A1 before A0).  Host only.
// ------------------------------------------------------------------------------------------------------
static void fp_ext_to_be32(const u32* ext_words, uint8_t* out) {
    u32 t[8], c[8];
    memcpy(t, ext_words, 32);
    fe_pack<FpCfg>(fe_canonical<FpCfg>(fe_mul<FpCfg>(fe_unpack<FpCfg>(t), fe_small<FpCfg>(32))), c);   // x*2^256 -> x
    for (int i = 0; i < 8; ++i) {
        const u32 w = c[7 - i];
        out[4 * i] = (uint8_t)(w >> 24); out[4 * i + 1] = (uint8_t)(w >> 16); out[4 * i + 2] = (uint8_t)(w >> 8); out[4 * i + 3] = (uint8_t)w;
    }
}
extern "C" int32_t dvn_groth16_proof_abi_encode(const uint64_t proof[32], const uint64_t commitment[8], const uint64_t commitment_pok[8],
                                                uint8_t out[384]) {
    if (!proof || !out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    const u32* p = (const u32*)proof;
    const u32* ar = p; const u32* bs = p + 16; const u32* krs = p + 48;
    fp_ext_to_be32(ar, out); fp_ext_to_be32(ar + 8, out + 32);
    fp_ext_to_be32(bs + 8, out + 64);  fp_ext_to_be32(bs, out + 96);          // Bs.X.A1, Bs.X.A0
    fp_ext_to_be32(bs + 24, out + 128); fp_ext_to_be32(bs + 16, out + 160);   // Bs.Y.A1, Bs.Y.A0
    fp_ext_to_be32(krs, out + 192); fp_ext_to_be32(krs + 8, out + 224);
    memset(out + 256, 0, 128);
    if (commitment) { fp_ext_to_be32((const u32*)commitment, out + 256); fp_ext_to_be32((const u32*)commitment + 8, out + 288); }
    if (commitment_pok) { fp_ext_to_be32((const u32*)commitment_pok, out + 320); fp_ext_to_be32((const u32*)commitment_pok + 8, out + 352); }
    return DVN_OK;
}


// ------------------------------------------------------------------------------------------------------
// Pedersen commitment + proof of knowledge (gnark-crypto ecc/bn254/fr/pedersen ProvingKey.Commit /
// ProveKnowledge, called from gnark's BSB22 commitment hint inside groth16.Prove): two G1 MSMs over the committed
// wire values, against Basis and BasisExpSigma.  With a single commitment (statetransition has exactly one,
// /root/reference/solidity/solidity.go:57-65) BatchProve's fold is the identity, so pok_out IS CommitmentPok.
// Outputs are G1Affine images (8 x uint64, infinity = zeros).  Both MSMs are in flight together (two lanes).
// ------------------------------------------------------------------------------------------------------
extern "C" int32_t dvn_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma,
                                       const uint64_t* values_mont, size_t n, uint32_t flags,
                                       uint64_t commitment_out[8], uint64_t pok_out[8]) {
    TRY(enter(ctx));
    if (!basis || !basis_exp_sigma || (!values_mont && n) || !commitment_out || !pok_out) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    if (basis->curve != 1 || basis_exp_sigma->curve != 1) return fail(DVN_ERR_INVALID_ARG, "bases must be G1 point sets");
    if (n > basis->n || n > basis_exp_sigma->n) return fail(DVN_ERR_INVALID_ARG, "len(values) exceeds the basis");
    std::lock_guard<std::mutex> lk(ctx->mu);
    u32* d_vals = nullptr;
    const u32* vals = (const u32*)values_mont;
    if (!(flags & 1) && n) {               // host values: one upload shared by both MSMs
        HIP_TRY(hipMalloc((void**)&d_vals, n * 32));
        hipError_t e = hipMemcpy(d_vals, values_mont, n * 32, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d_vals); (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "upload failed: %s", hipGetErrorString(e)); }
        vals = d_vals;
    }
    typedef dvn_host::HG1 H1;
    MsmLane *la = nullptr, *lb = nullptr;
    H1::X c, k;
    int32_t rc = dvn_internal_msm_submit_g1(ctx, basis, 0, n, vals, &la);
    if (rc == DVN_OK) rc = dvn_internal_msm_submit_g1(ctx, basis_exp_sigma, 0, n, vals, &lb);
    if (rc == DVN_OK) rc = dvn_internal_msm_collect_g1(ctx, la, &c); else if (la) { H1::X t; (void)dvn_internal_msm_collect_g1(ctx, la, &t); }
    if (rc == DVN_OK) rc = dvn_internal_msm_collect_g1(ctx, lb, &k);
    if (d_vals) (void)hipFree(d_vals);
    if (rc != DVN_OK) return rc;
    uint64_t j[12];
    H1::write_jac(c, j); memcpy(commitment_out, j, 64); if (H1::is_inf(c)) memset(commitment_out, 0, 64);
    H1::write_jac(k, j); memcpy(pok_out, j, 64); if (H1::is_inf(k)) memset(pok_out, 0, 64);
    return DVN_OK;
}
