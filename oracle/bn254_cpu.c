/* CPU oracle for the davinci-node prover hot path -- TEST INFRASTRUCTURE, not product.
 *
 * A plain-C restatement of what prover/prover_cpu.go:37 (groth16.Prove) spends its time in, following the
 * published algorithms of the un-vendored modules pinned in /root/reference/go.mod:15-16:
 *   - gnark-crypto ecc/bn254 multiexp.go  -> orc_msm_g1 / orc_msm_g2
 *   - gnark-crypto ecc/bn254/fr/fft       -> orc_ntt
 *   - gnark backend/groth16/bn254 prove.go computeH -> orc_compute_h
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PARITY UNPINNED by reference vectors (none exist); pinned against oracle/bn254_py.py and
 * the reference's curve constants by tests/test_oracle.py.
 *
 * All field elements cross this API as 4 x u64 little-endian Montgomery limbs unless a parameter
 * says "canonical".
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
#include "bn254_field.h"

#define FE fe
#define F(op) fp_##op
#define ECN(x) g1_##x
#include "bn254_ec_tmpl.h"
#undef FE
#undef F
#undef ECN

#define FE fe2
#define F(op) fp2_##op
#define ECN(x) g2_##x
#include "bn254_ec_tmpl.h"
#undef FE
#undef F
#undef ECN

/* ------------------------------------------------------------------ deterministic inputs (bn254_py.py) */
static inline u64 splitmix_at(u64 seed, u64 index) {
    u64 z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* 256-bit value mod r: the value is < 2^256 < 6r, so at most 5 subtractions */
static inline void reduce_mod_r(u64 v[4]) { while (geq4(v, FR_MOD)) sub4(v, v, FR_MOD); }
static inline void fr_at_canonical(u64 seed, u64 i, u64 out[4]) {
    for (int k = 0; k < 4; ++k) out[k] = splitmix_at(seed, 4 * i + k);
    reduce_mod_r(out);
}
static inline void witness_like_canonical(u64 seed, u64 i, u64 out[4]) {
    u64 sel = splitmix_at(seed ^ 0x5EED5EED5EED5EEDULL, i) % 100;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (sel < 40) return;
    if (sel < 65) { out[0] = 1; return; }
    if (sel < 90) { out[0] = splitmix_at(seed, 4 * i); return; }
    fr_at_canonical(seed, i, out);
}

/* kind 0 = uniform, 1 = witness-like; output Montgomery */
void orc_scalars(u64 seed, int kind, size_t n, u64* out_mont) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        fe c;
        if (kind == 0) fr_at_canonical(seed, i, c.l); else witness_like_canonical(seed, i, c.l);
        fe m; fr_to_mont(&m, &c);
        memcpy(out_mont + 4 * i, m.l, 32);
    }
}

/* ------------------------------------------------------------------ field KAT entry points */
void orc_fp_mul(const u64* a, const u64* b, u64* r) { fp_mul((fe*)r, (const fe*)a, (const fe*)b); }
void orc_fr_mul(const u64* a, const u64* b, u64* r) { fr_mul((fe*)r, (const fe*)a, (const fe*)b); }
void orc_fp_add(const u64* a, const u64* b, u64* r) { fp_add((fe*)r, (const fe*)a, (const fe*)b); }
void orc_fp_sub(const u64* a, const u64* b, u64* r) { fp_sub((fe*)r, (const fe*)a, (const fe*)b); }
void orc_fr_add(const u64* a, const u64* b, u64* r) { fr_add((fe*)r, (const fe*)a, (const fe*)b); }
void orc_fr_sub(const u64* a, const u64* b, u64* r) { fr_sub((fe*)r, (const fe*)a, (const fe*)b); }
void orc_fp_inv(const u64* a, u64* r) { fp_inv((fe*)r, (const fe*)a); }
void orc_fr_inv(const u64* a, u64* r) { fr_inv((fe*)r, (const fe*)a); }
void orc_fp_to_mont(const u64* a, u64* r) { fp_to_mont((fe*)r, (const fe*)a); }
void orc_fp_from_mont(const u64* a, u64* r) { fp_from_mont((fe*)r, (const fe*)a); }
void orc_fr_to_mont(const u64* a, u64* r) { fr_to_mont((fe*)r, (const fe*)a); }
void orc_fr_from_mont(const u64* a, u64* r) { fr_from_mont((fe*)r, (const fe*)a); }
void orc_fp2_mul(const u64* a, const u64* b, u64* r) { fp2_mul((fe2*)r, (const fe2*)a, (const fe2*)b); }
void orc_fp2_inv(const u64* a, u64* r) { fp2_inv((fe2*)r, (const fe2*)a); }
void orc_fr_vec_from_mont(const u64* in, u64* out, size_t n) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) fr_from_mont((fe*)(out + 4 * i), (const fe*)(in + 4 * i));
}

/* ------------------------------------------------------------------ group entry points */
static const u64 G2_GEN_MONT[16] = {
    0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL,
    0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL,
    0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL,
    0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL};

void orc_g1_generator(u64* out8) {
    fe one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    fp_to_mont((fe*)out8, &one); fp_to_mont((fe*)(out8 + 4), &two);
}
void orc_g2_generator(u64* out16) { memcpy(out16, G2_GEN_MONT, 128); }

int orc_g1_on_curve(const u64* p8) {
    const g1_aff* p = (const g1_aff*)p8;
    if (g1_aff_is_inf(p)) return 1;
    fe l, r, b = {{3, 0, 0, 0}}; fp_to_mont(&b, &b);
    fp_sqr(&l, &p->y); fp_sqr(&r, &p->x); fp_mul(&r, &r, &p->x); fp_add(&r, &r, &b);
    return fe_eq(&l, &r);
}
int orc_g2_on_curve(const u64* p16) {
    const g2_aff* p = (const g2_aff*)p16;
    if (g2_aff_is_inf(p)) return 1;
    /* b' = 3/(9+u) */
    fe2 nine_u, b3, binv, bt, l, r;
    fe nine = {{9, 0, 0, 0}}, one = {{1, 0, 0, 0}}, three = {{3, 0, 0, 0}};
    fp_to_mont(&nine_u.a0, &nine); fp_to_mont(&nine_u.a1, &one);
    fp_to_mont(&b3.a0, &three); fp_set_zero(&b3.a1);
    fp2_inv(&binv, &nine_u); fp2_mul(&bt, &b3, &binv);
    fp2_sqr(&l, &p->y); fp2_sqr(&r, &p->x); fp2_mul(&r, &r, &p->x); fp2_add(&r, &r, &bt);
    return fp2_eq(&l, &r);
}
/* out = a + b (affine, any of them may be infinity) */
void orc_g1_add(const u64* a8, const u64* b8, u64* out8) {
    g1_xyzz t; g1_xyzz_from_aff(&t, (const g1_aff*)a8); g1_xyzz_add_aff(&t, (const g1_aff*)b8, 0);
    g1_xyzz_to_aff((g1_aff*)out8, &t);
}
void orc_g2_add(const u64* a16, const u64* b16, u64* out16) {
    g2_xyzz t; g2_xyzz_from_aff(&t, (const g2_aff*)a16); g2_xyzz_add_aff(&t, (const g2_aff*)b16, 0);
    g2_xyzz_to_aff((g2_aff*)out16, &t);
}
/* out = [k]p, k canonical 4 x u64 */
void orc_g1_mul(const u64* p8, const u64* k4, u64* out8) {
    g1_xyzz t; g1_scalar_mul(&t, (const g1_aff*)p8, k4); g1_xyzz_to_aff((g1_aff*)out8, &t);
}
void orc_g2_mul(const u64* p16, const u64* k4, u64* out16) {
    g2_xyzz t; g2_scalar_mul(&t, (const g2_aff*)p16, k4); g2_xyzz_to_aff((g2_aff*)out16, &t);
}
/* normalise an extended-Jacobian (X,Y,ZZ,ZZZ) or Jacobian (X,Y,Z) point as returned by the device library */
void orc_g1_xyzz_to_affine(const u64* in16, u64* out8) { g1_xyzz_to_aff((g1_aff*)out8, (const g1_xyzz*)in16); }
void orc_g2_xyzz_to_affine(const u64* in32, u64* out16) { g2_xyzz_to_aff((g2_aff*)out16, (const g2_xyzz*)in32); }
void orc_g1_jac_to_affine(const u64* in12, u64* out8) {
    const fe* X = (const fe*)in12; const fe* Y = X + 1; const fe* Z = X + 2;
    g1_aff* o = (g1_aff*)out8;
    if (fe_is_zero(Z)) { memset(out8, 0, 64); return; }
    fe zi, zi2, zi3; fp_inv(&zi, Z); fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
    fp_mul(&o->x, X, &zi2); fp_mul(&o->y, Y, &zi3);
}
void orc_g2_jac_to_affine(const u64* in24, u64* out16) {
    const fe2* X = (const fe2*)in24; const fe2* Y = X + 1; const fe2* Z = X + 2;
    g2_aff* o = (g2_aff*)out16;
    if (fp2_is_zero(Z)) { memset(out16, 0, 128); return; }
    fe2 zi, zi2, zi3; fp2_inv(&zi, Z); fp2_sqr(&zi2, &zi); fp2_mul(&zi3, &zi2, &zi);
    fp2_mul(&o->x, X, &zi2); fp2_mul(&o->y, Y, &zi3);
}

/* P_i = [s0]G + i*[s1]G, (s0, s1) = first two Fr of stream `seed` (bn254_py.synthetic_g1_points) */
void orc_g1_synthetic_points(u64 seed, size_t n, u64* out, int nthreads) {
    u64 s0[4], s1[4], g[8]; fr_at_canonical(seed, 0, s0); fr_at_canonical(seed, 1, s1);
    orc_g1_generator(g);
    g1_aff base, step; orc_g1_mul(g, s0, (u64*)&base); orc_g1_mul(g, s1, (u64*)&step);
    g1_arith_progression((g1_aff*)out, &base, &step, n, nthreads > 0 ? nthreads : omp_get_max_threads());
}
void orc_g2_synthetic_points(u64 seed, size_t n, u64* out, int nthreads) {
    u64 s0[4], s1[4]; fr_at_canonical(seed, 0, s0); fr_at_canonical(seed, 1, s1);
    g2_aff base, step; orc_g2_mul(G2_GEN_MONT, s0, (u64*)&base); orc_g2_mul(G2_GEN_MONT, s1, (u64*)&step);
    g2_arith_progression((g2_aff*)out, &base, &step, n, nthreads > 0 ? nthreads : omp_get_max_threads());
}

/* MultiExp: points affine Montgomery, scalars Montgomery (as gnark passes them); out = affine Montgomery.
 * returns 0, or -1 on the MultiExp error condition len(points) != len(scalars) (checked by the caller
 * passing both lengths). */
int orc_msm_g1(const u64* points, size_t n_points, const u64* scalars_mont, size_t n_scalars, int nthreads, u64* out8) {
    if (n_points != n_scalars) return -1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    u64* canon = (u64*)malloc(32 * (n_scalars ? n_scalars : 1));
    orc_fr_vec_from_mont(scalars_mont, canon, n_scalars);
    g1_xyzz r; g1_msm(&r, (const g1_aff*)points, canon, n_points, nthreads);
    g1_xyzz_to_aff((g1_aff*)out8, &r);
    free(canon);
    return 0;
}
/* per-window sums of the signed-digit Pippenger with window size c: out = (254/c + 1) x 16 u64, extended
 * Jacobian (X, Y, ZZ, ZZZ) Montgomery, infinity = all zero.  Used to test the window-sharded multi-GPU path. */
int orc_msm_g1_window_sums(const u64* points, const u64* scalars_mont, size_t n, int c, int nthreads, u64* out) {
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    int nwin = 254 / c + 1;
    u64* canon = (u64*)malloc(32 * (n ? n : 1));
    orc_fr_vec_from_mont(scalars_mont, canon, n);
    g1_xyzz r; g1_xyzz* ws = (g1_xyzz*)calloc(nwin, sizeof(g1_xyzz));
    for (int w = 0; w < nwin; ++w) g1_xyzz_set_inf(&ws[w]);
    g1_msm_ex(&r, (const g1_aff*)points, canon, n, nthreads, c, ws);
    for (int w = 0; w < nwin; ++w) {
        if (g1_xyzz_is_inf(&ws[w])) memset(out + 16 * w, 0, 128); else memcpy(out + 16 * w, &ws[w], 128);
    }
    free(ws); free(canon);
    return nwin;
}
int orc_msm_g2(const u64* points, size_t n_points, const u64* scalars_mont, size_t n_scalars, int nthreads, u64* out16) {
    if (n_points != n_scalars) return -1;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    u64* canon = (u64*)malloc(32 * (n_scalars ? n_scalars : 1));
    orc_fr_vec_from_mont(scalars_mont, canon, n_scalars);
    g2_xyzz r; g2_msm(&r, (const g2_aff*)points, canon, n_points, nthreads);
    g2_xyzz_to_aff((g2_aff*)out16, &r);
    free(canon);
    return 0;
}

/* ------------------------------------------------------------------ fr/fft Domain */
static const u64 FR_ROOT_2_28_MONT[4] = {0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL, 0x1860ef942963f9e7ULL};

typedef struct {
    int log_n; size_t n;
    fe gen, gen_inv, card_inv, coset, coset_inv;
    fe* tw;      /* gen^j,     j < n/2 */
    fe* tw_inv;  /* gen_inv^j, j < n/2 */
} domain_t;

static void fill_powers(fe* out, const fe* base, size_t count) {
    /* out[j] = base^j; chunked so threads start from base^(chunk start) */
    if (count == 0) return;
    const size_t CH = 1 << 14;
    size_t nch = (count + CH - 1) / CH;
    #pragma omp parallel for schedule(static)
    for (size_t ch = 0; ch < nch; ++ch) {
        size_t lo = ch * CH, hi = lo + CH < count ? lo + CH : count;
        fe cur; fr_pow_u64(&cur, base, lo);
        for (size_t j = lo; j < hi; ++j) { out[j] = cur; fr_mul(&cur, &cur, base); }
    }
}

static domain_t* domain_new(int log_n) {
    domain_t* d = (domain_t*)calloc(1, sizeof(domain_t));
    d->log_n = log_n; d->n = (size_t)1 << log_n;
    memcpy(d->gen.l, FR_ROOT_2_28_MONT, 32);
    for (int i = log_n; i < 28; ++i) fr_mul(&d->gen, &d->gen, &d->gen);
    fr_inv(&d->gen_inv, &d->gen);
    fe nn = {{(u64)d->n, 0, 0, 0}}; fr_to_mont(&nn, &nn); fr_inv(&d->card_inv, &nn);
    fe five = {{5, 0, 0, 0}}; fr_to_mont(&d->coset, &five); fr_inv(&d->coset_inv, &d->coset);
    size_t half = d->n / 2 ? d->n / 2 : 1;
    d->tw = (fe*)malloc(sizeof(fe) * half); d->tw_inv = (fe*)malloc(sizeof(fe) * half);
    fill_powers(d->tw, &d->gen, half); fill_powers(d->tw_inv, &d->gen_inv, half);
    return d;
}
static void domain_free(domain_t* d) { free(d->tw); free(d->tw_inv); free(d); }

static inline size_t bitrev_sz(size_t i, int log_n) {
    size_t r = 0;
    for (int b = 0; b < log_n; ++b) { r = (r << 1) | ((i >> b) & 1); }
    return r;
}

/* DIF: natural in -> bit-reversed out; butterfly (a,b) -> (a+b, (a-b)*w) */
static void dif_fft(fe* a, const domain_t* d, const fe* tw) {
    size_t n = d->n;
    for (int s = 0; s < d->log_n; ++s) {
        size_t half = n >> (s + 1);
        #pragma omp parallel for schedule(static)
        for (size_t idx = 0; idx < n / 2; ++idx) {
            size_t blk = idx / half, j = idx % half, i0 = blk * 2 * half + j, i1 = i0 + half;
            fe x = a[i0], y = a[i1], t;
            fr_add(&a[i0], &x, &y); fr_sub(&t, &x, &y); fr_mul(&a[i1], &t, &tw[j << s]);
        }
    }
}
/* DIT: bit-reversed in -> natural out; butterfly (a,b) -> (a+b*w, a-b*w) */
static void dit_fft(fe* a, const domain_t* d, const fe* tw) {
    size_t n = d->n;
    for (int s = d->log_n - 1; s >= 0; --s) {
        size_t half = n >> (s + 1);
        #pragma omp parallel for schedule(static)
        for (size_t idx = 0; idx < n / 2; ++idx) {
            size_t blk = idx / half, j = idx % half, i0 = blk * 2 * half + j, i1 = i0 + half;
            fe x = a[i0], y; fr_mul(&y, &a[i1], &tw[j << s]);
            fr_add(&a[i0], &x, &y); fr_sub(&a[i1], &x, &y);
        }
    }
}
static void scale_by_powers(fe* a, const domain_t* d, const fe* base, const fe* extra, int bitrev_index) {
    /* a[i] *= base^(bitrev_index ? rev(i) : i) * (extra ? *extra : 1) */
    size_t n = d->n;
    fe* pw = (fe*)malloc(sizeof(fe) * n);
    fill_powers(pw, base, n);
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        size_t k = bitrev_index ? bitrev_sz(i, d->log_n) : i;
        fe s = pw[k];
        if (extra) fr_mul(&s, &s, extra);
        fr_mul(&a[i], &a[i], &s);
    }
    free(pw);
}
static void domain_fft(fe* a, const domain_t* d, int decimation /*0 DIF, 1 DIT*/, int coset) {
    if (coset) scale_by_powers(a, d, &d->coset, NULL, decimation == 1);
    if (decimation == 0) dif_fft(a, d, d->tw); else dit_fft(a, d, d->tw);
}
static void domain_fft_inverse(fe* a, const domain_t* d, int decimation, int coset) {
    if (decimation == 0) dif_fft(a, d, d->tw_inv); else dit_fft(a, d, d->tw_inv);
    if (!coset) {
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < d->n; ++i) fr_mul(&a[i], &a[i], &d->card_inv);
    } else {
        scale_by_powers(a, d, &d->coset_inv, &d->card_inv, decimation == 0);
    }
}

/* gnark fft.Domain.FFT / FFTInverse on n = 2^log_n Montgomery elements, in place */
int orc_ntt(u64* data, int log_n, int inverse, int decimation, int coset, int nthreads) {
    if (log_n < 0 || log_n > 28) return -1;
    if (nthreads > 0) omp_set_num_threads(nthreads);
    domain_t* d = domain_new(log_n);
    if (inverse) domain_fft_inverse((fe*)data, d, decimation, coset); else domain_fft((fe*)data, d, decimation, coset);
    domain_free(d);
    return 0;
}
/* timing variant: domain construction excluded, `reps` forward(DIF)+inverse(DIT) round trips */
double orc_ntt_roundtrip_seconds(u64* data, int log_n, int reps, int nthreads) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    domain_t* d = domain_new(log_n);
    double t0 = omp_get_wtime();
    for (int r = 0; r < reps; ++r) { domain_fft((fe*)data, d, 0, 0); domain_fft_inverse((fe*)data, d, 1, 0); }
    double t1 = omp_get_wtime();
    domain_free(d);
    return t1 - t0;
}

/* gnark groth16 computeH: a, b, c have `len` <= n elements; h_out gets n elements (bit-reversed) */
int orc_compute_h(const u64* a_in, const u64* b_in, const u64* c_in, size_t len, int log_n, u64* h_out, int nthreads) {
    size_t n = (size_t)1 << log_n;
    if (len > n) return -1;
    if (nthreads > 0) omp_set_num_threads(nthreads);
    domain_t* d = domain_new(log_n);
    fe* a = (fe*)h_out;
    fe* b = (fe*)calloc(n, sizeof(fe));
    fe* c = (fe*)calloc(n, sizeof(fe));
    memset(a, 0, n * sizeof(fe));
    memcpy(a, a_in, len * 32); memcpy(b, b_in, len * 32); memcpy(c, c_in, len * 32);
    domain_fft_inverse(a, d, 0, 0); domain_fft_inverse(b, d, 0, 0); domain_fft_inverse(c, d, 0, 0);
    domain_fft(a, d, 1, 1); domain_fft(b, d, 1, 1); domain_fft(c, d, 1, 1);
    fe den, one; fr_set_one(&one);
    fr_pow_u64(&den, &d->coset, (u64)n); fr_sub(&den, &den, &one); fr_inv(&den, &den);
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        fe t; fr_mul(&t, &a[i], &b[i]); fr_sub(&t, &t, &c[i]); fr_mul(&a[i], &t, &den);
    }
    domain_fft_inverse(a, d, 0, 1);
    free(b); free(c); domain_free(d);
    return 0;
}

int orc_num_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------ synthetic Groth16 checker */
/* Restates oracle/bn254_py.py SyntheticR1CS + groth16_setup's trapdoor + groth16_expected_from_trapdoor:
 * the proof a correct prover must output for injected (r, s), from O(n) field work and three fixed-base
 * scalar multiplications -- no MSM, no NTT, so it is independent of everything it checks. */
typedef struct { u64 s; } smix;
static inline u64 smix_next(smix* g) {
    g->s += 0x9E3779B97F4A7C15ULL;
    u64 z = g->s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline void smix_next_fr_mont(smix* g, fe* out) {
    fe c; for (int k = 0; k < 4; ++k) c.l[k] = smix_next(g);
    reduce_mod_r(c.l); fr_to_mont(out, &c);
}
static inline void fr_from_u64_mont(fe* r, u64 v) { fe c = {{v, 0, 0, 0}}; fr_to_mont(r, &c); }

int orc_groth16_synthetic_expected(u64 cs_seed, uint32_t n_constraints, uint32_t n_public, u64 setup_seed,
                                   const u64* r_mont, const u64* s_mont, u64* witness_out, u64* proof_out) {
    const uint32_t n_free = 4;
    const size_t m = (size_t)n_public + n_free + n_constraints;
    fe* w = (fe*)malloc(sizeof(fe) * m);
    uint32_t* l1 = (uint32_t*)malloc(4 * (size_t)n_constraints); uint32_t* l2 = (uint32_t*)malloc(4 * (size_t)n_constraints);
    uint32_t* rr = (uint32_t*)malloc(4 * (size_t)n_constraints); uint32_t* kk = (uint32_t*)malloc(4 * (size_t)n_constraints);
    smix g = {cs_seed};
    size_t cur = 0;
    fr_set_one(&w[cur++]);
    for (uint32_t i = 1; i < n_public; ++i) smix_next_fr_mont(&g, &w[cur++]);
    for (uint32_t i = 0; i < n_free; ++i) smix_next_fr_mont(&g, &w[cur++]);
    for (uint32_t j = 0; j < n_constraints; ++j) {
        l1[j] = (uint32_t)(smix_next(&g) % cur); l2[j] = (uint32_t)(smix_next(&g) % cur);
        kk[j] = (uint32_t)(smix_next(&g) % 7 + 1); rr[j] = (uint32_t)(smix_next(&g) % cur);
        fe kf, t; fr_from_u64_mont(&kf, kk[j]);
        fr_mul(&t, &kf, &w[l2[j]]); fr_add(&t, &t, &w[l1[j]]); fr_mul(&w[cur], &t, &w[rr[j]]);
        ++cur;
    }
    if (witness_out) memcpy(witness_out, w, sizeof(fe) * m);
    /* trapdoor */
    smix gs = {setup_seed};
    fe tw[5], one; fr_set_one(&one);
    for (int i = 0; i < 5; ++i) { smix_next_fr_mont(&gs, &tw[i]); if (fe_is_zero(&tw[i])) tw[i] = one; }
    const fe tau = tw[0], alpha = tw[1], beta = tw[2], delta = tw[4];
    int log_n = 1; while (((size_t)1 << log_n) < n_constraints) ++log_n;
    const size_t n = (size_t)1 << log_n;
    fe omega; memcpy(omega.l, FR_ROOT_2_28_MONT, 32);
    for (int i = log_n; i < 28; ++i) fr_mul(&omega, &omega, &omega);
    /* L_j(tau), j < n_constraints is all that the rows touch */
    fe ztau, nn, zn; fr_pow_u64(&ztau, &tau, (u64)n); fr_sub(&ztau, &ztau, &one);
    fr_from_u64_mont(&nn, (u64)n); fr_inv(&nn, &nn); fr_mul(&zn, &ztau, &nn);
    fe* wj = (fe*)malloc(sizeof(fe) * n_constraints); fe* den = (fe*)malloc(sizeof(fe) * n_constraints);
    fe* pre = (fe*)malloc(sizeof(fe) * n_constraints); fe* L = (fe*)malloc(sizeof(fe) * n_constraints);
    fe curw = one, acc = one;
    for (uint32_t j = 0; j < n_constraints; ++j) { wj[j] = curw; fr_sub(&den[j], &tau, &curw); pre[j] = acc; fr_mul(&acc, &acc, &den[j]); fr_mul(&curw, &curw, &omega); }
    fe inv; fr_inv(&inv, &acc);
    for (uint32_t j = n_constraints; j-- > 0;) {
        fe di; fr_mul(&di, &inv, &pre[j]); fr_mul(&inv, &inv, &den[j]);
        fr_mul(&L[j], &zn, &wj[j]); fr_mul(&L[j], &L[j], &di);
    }
    fe* At = (fe*)calloc(m, sizeof(fe)); fe* Bt = (fe*)calloc(m, sizeof(fe)); fe* Ct = (fe*)calloc(m, sizeof(fe));
    for (uint32_t j = 0; j < n_constraints; ++j) {
        fe kf, t;
        if (l1[j] != l2[j]) { fr_add(&At[l1[j]], &At[l1[j]], &L[j]); fr_from_u64_mont(&kf, kk[j]); fr_mul(&t, &kf, &L[j]); fr_add(&At[l2[j]], &At[l2[j]], &t); }
        else { fr_from_u64_mont(&kf, 1 + (u64)kk[j]); fr_mul(&t, &kf, &L[j]); fr_add(&At[l1[j]], &At[l1[j]], &t); }
        fr_add(&Bt[rr[j]], &Bt[rr[j]], &L[j]);
        size_t o = (size_t)n_public + n_free + j;
        fr_add(&Ct[o], &Ct[o], &L[j]);
    }
    fe a_tau = {{0}}, b_tau = {{0}}, c_tau = {{0}}, priv = {{0}}, dinv, t;
    fr_inv(&dinv, &delta);
    for (size_t i = 0; i < m; ++i) {
        fr_mul(&t, &w[i], &At[i]); fr_add(&a_tau, &a_tau, &t);
        fr_mul(&t, &w[i], &Bt[i]); fr_add(&b_tau, &b_tau, &t);
        fr_mul(&t, &w[i], &Ct[i]); fr_add(&c_tau, &c_tau, &t);
        if (i >= n_public) {
            fe k1, k2; fr_mul(&k1, &beta, &At[i]); fr_mul(&k2, &alpha, &Bt[i]); fr_add(&k1, &k1, &k2); fr_add(&k1, &k1, &Ct[i]);
            fr_mul(&k1, &k1, &dinv); fr_mul(&k1, &k1, &w[i]); fr_add(&priv, &priv, &k1);
        }
    }
    fe r, s, ar, bs, krs, h_tau, zi;
    memcpy(r.l, r_mont, 32); memcpy(s.l, s_mont, 32);
    fr_inv(&zi, &ztau);
    fr_mul(&h_tau, &a_tau, &b_tau); fr_sub(&h_tau, &h_tau, &c_tau); fr_mul(&h_tau, &h_tau, &zi);
    fr_mul(&t, &r, &delta); fr_add(&ar, &alpha, &a_tau); fr_add(&ar, &ar, &t);
    fr_mul(&t, &s, &delta); fr_add(&bs, &beta, &b_tau); fr_add(&bs, &bs, &t);
    fr_mul(&krs, &h_tau, &ztau); fr_mul(&krs, &krs, &dinv); fr_add(&krs, &krs, &priv);
    fr_mul(&t, &s, &ar); fr_add(&krs, &krs, &t);
    fr_mul(&t, &r, &bs); fr_add(&krs, &krs, &t);
    fr_mul(&t, &r, &s); fr_mul(&t, &t, &delta); fr_sub(&krs, &krs, &t);
    fe arc, bsc, krsc; fr_from_mont(&arc, &ar); fr_from_mont(&bsc, &bs); fr_from_mont(&krsc, &krs);
    u64 gen1[8]; orc_g1_generator(gen1);
    orc_g1_mul(gen1, arc.l, proof_out);
    orc_g2_mul(G2_GEN_MONT, bsc.l, proof_out + 8);
    orc_g1_mul(gen1, krsc.l, proof_out + 24);
    free(w); free(l1); free(l2); free(rr); free(kk); free(wj); free(den); free(pre); free(L); free(At); free(Bt); free(Ct);
    return (int)m;
}
