/* CPU restatement of gnark-crypto's BN254 field arithmetic -- TEST INFRASTRUCTURE (oracle), not product.
 *
 * Restates ecc/bn254/fp and ecc/bn254/fr of github.com/consensys/gnark-crypto
 * v0.19.3-0.20260126145145-b5cf053fbc34 (/root/reference/go.mod:16, un-vendored): Element = [4]uint64
 * little-endian limbs in Montgomery form with R = 2^256.  Portable C with
 * unsigned __int128 instead of gnark-crypto's amd64 assembly.  Checked against oracle/bn254_py.py and the
 * reference constants of config/statetransition_vkey.sol:42-59 by tests/test_oracle.py.
 * PARITY UNPINNED beyond those constants -- see oracle/bn254_py.py header.
 */
#ifndef ORACLE_BN254_FIELD_H
#define ORACLE_BN254_FIELD_H
#include <stdint.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef struct { u64 l[4]; } fe;

static const u64 FP_MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 FP_ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
static const u64 FP_R2[4]  = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
#define FP_INV 0x87d20782e4866389ULL
static const u64 FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 FR_ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
static const u64 FR_R2[4]  = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
#define FR_INV 0xc2e1f593efffffffULL

static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int geq4(const u64* a, const u64* p) {
    for (int i = 3; i >= 0; --i) { if (a[i] > p[i]) return 1; if (a[i] < p[i]) return 0; }
    return 1;
}
static inline void sub4(u64* r, const u64* a, const u64* b) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - br; r[i] = (u64)d; br = (d >> 64) & 1; }
}
static inline void mod_add(fe* r, const fe* a, const fe* b, const u64* p) {
    u128 c = 0; u64 t[4];
    for (int i = 0; i < 4; ++i) { c += (u128)a->l[i] + b->l[i]; t[i] = (u64)c; c >>= 64; }
    if (c || geq4(t, p)) sub4(t, t, p);
    memcpy(r->l, t, 32);
}
static inline void mod_sub(fe* r, const fe* a, const fe* b, const u64* p) {
    u128 br = 0; u64 t[4];
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a->l[i] - b->l[i] - br; t[i] = (u64)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)t[i] + p[i]; t[i] = (u64)c; c >>= 64; } }
    memcpy(r->l, t, 32);
}
static inline void mod_neg(fe* r, const fe* a, const u64* p) {
    if (fe_is_zero(a)) { *r = *a; return; }
    sub4(r->l, p, a->l);
}
/* CIOS Montgomery multiplication, R = 2^256 */
static inline void mont_mul(fe* r, const fe* a, const fe* b, const u64* p, u64 inv) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        u64 m = t[0] * inv;
        c = (u128)m * p[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * p[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq4(t, p)) sub4(t, t, p);
    memcpy(r->l, t, 32);
}

/* ---- Fp ---- */
static inline void fp_add(fe* r, const fe* a, const fe* b) { mod_add(r, a, b, FP_MOD); }
static inline void fp_sub(fe* r, const fe* a, const fe* b) { mod_sub(r, a, b, FP_MOD); }
static inline void fp_neg(fe* r, const fe* a) { mod_neg(r, a, FP_MOD); }
static inline void fp_mul(fe* r, const fe* a, const fe* b) { mont_mul(r, a, b, FP_MOD, FP_INV); }
static inline void fp_sqr(fe* r, const fe* a) { mont_mul(r, a, a, FP_MOD, FP_INV); }
static inline void fp_dbl(fe* r, const fe* a) { mod_add(r, a, a, FP_MOD); }
static inline void fp_set_one(fe* r) { memcpy(r->l, FP_ONE, 32); }
static inline void fp_set_zero(fe* r) { memset(r->l, 0, 32); }
static inline int fp_is_zero(const fe* a) { return fe_is_zero(a); }
static inline int fp_eq(const fe* a, const fe* b) { return fe_eq(a, b); }
/* ---- Fr ---- */
static inline void fr_add(fe* r, const fe* a, const fe* b) { mod_add(r, a, b, FR_MOD); }
static inline void fr_sub(fe* r, const fe* a, const fe* b) { mod_sub(r, a, b, FR_MOD); }
static inline void fr_mul(fe* r, const fe* a, const fe* b) { mont_mul(r, a, b, FR_MOD, FR_INV); }
static inline void fr_set_one(fe* r) { memcpy(r->l, FR_ONE, 32); }
static inline void fr_to_mont(fe* r, const fe* a) { fe r2; memcpy(r2.l, FR_R2, 32); fr_mul(r, a, &r2); }
static inline void fr_from_mont(fe* r, const fe* a) { fe one = {{1, 0, 0, 0}}; fr_mul(r, a, &one); }
static inline void fp_to_mont(fe* r, const fe* a) { fe r2; memcpy(r2.l, FP_R2, 32); fp_mul(r, a, &r2); }
static inline void fp_from_mont(fe* r, const fe* a) { fe one = {{1, 0, 0, 0}}; fp_mul(r, a, &one); }

/* x^e for a 256-bit exponent e (little-endian limbs), Montgomery in/out */
static inline void mont_pow(fe* r, const fe* x, const u64* e, const u64* p, u64 inv, const u64* one) {
    fe acc; memcpy(acc.l, one, 32);
    for (int i = 255; i >= 0; --i) {
        mont_mul(&acc, &acc, &acc, p, inv);
        if ((e[i >> 6] >> (i & 63)) & 1) mont_mul(&acc, &acc, x, p, inv);
    }
    *r = acc;
}
static inline void fp_inv(fe* r, const fe* a) {   /* Fermat: a^(p-2); 0 -> 0 like gnark's Inverse */
    u64 e[4]; u64 two[4] = {2, 0, 0, 0}; sub4(e, FP_MOD, two);
    mont_pow(r, a, e, FP_MOD, FP_INV, FP_ONE);
}
static inline void fr_inv(fe* r, const fe* a) {
    u64 e[4]; u64 two[4] = {2, 0, 0, 0}; sub4(e, FR_MOD, two);
    mont_pow(r, a, e, FR_MOD, FR_INV, FR_ONE);
}
static inline void fr_pow_u64(fe* r, const fe* x, u64 k) {
    u64 e[4] = {k, 0, 0, 0};
    mont_pow(r, x, e, FR_MOD, FR_INV, FR_ONE);
}

/* ---- Fp2 = Fp[u]/(u^2+1), element a0 + a1 u (gnark-crypto ecc/bn254/internal/fptower E2) ---- */
typedef struct { fe a0, a1; } fe2;
static inline void fp2_add(fe2* r, const fe2* a, const fe2* b) { fp_add(&r->a0, &a->a0, &b->a0); fp_add(&r->a1, &a->a1, &b->a1); }
static inline void fp2_sub(fe2* r, const fe2* a, const fe2* b) { fp_sub(&r->a0, &a->a0, &b->a0); fp_sub(&r->a1, &a->a1, &b->a1); }
static inline void fp2_neg(fe2* r, const fe2* a) { fp_neg(&r->a0, &a->a0); fp_neg(&r->a1, &a->a1); }
static inline void fp2_dbl(fe2* r, const fe2* a) { fp_dbl(&r->a0, &a->a0); fp_dbl(&r->a1, &a->a1); }
static inline void fp2_mul(fe2* r, const fe2* a, const fe2* b) {
    fe t0, t1, s0, s1, m;
    fp_mul(&t0, &a->a0, &b->a0); fp_mul(&t1, &a->a1, &b->a1);
    fp_add(&s0, &a->a0, &a->a1); fp_add(&s1, &b->a0, &b->a1); fp_mul(&m, &s0, &s1);
    fp_sub(&m, &m, &t0); fp_sub(&m, &m, &t1);
    fp_sub(&r->a0, &t0, &t1); r->a1 = m;
}
static inline void fp2_sqr(fe2* r, const fe2* a) {
    fe s, d, m;
    fp_add(&s, &a->a0, &a->a1); fp_sub(&d, &a->a0, &a->a1); fp_mul(&m, &a->a0, &a->a1);
    fp_mul(&r->a0, &s, &d); fp_dbl(&r->a1, &m);
}
static inline void fp2_inv(fe2* r, const fe2* a) {
    fe t0, t1; fp_sqr(&t0, &a->a0); fp_sqr(&t1, &a->a1); fp_add(&t0, &t0, &t1); fp_inv(&t0, &t0);
    fp_mul(&r->a0, &a->a0, &t0); fp_mul(&t1, &a->a1, &t0); fp_neg(&r->a1, &t1);
}
static inline void fp2_set_one(fe2* r) { fp_set_one(&r->a0); fp_set_zero(&r->a1); }
static inline void fp2_set_zero(fe2* r) { fp_set_zero(&r->a0); fp_set_zero(&r->a1); }
static inline int fp2_is_zero(const fe2* a) { return fe_is_zero(&a->a0) && fe_is_zero(&a->a1); }
static inline int fp2_eq(const fe2* a, const fe2* b) { return fe_eq(&a->a0, &b->a0) && fe_eq(&a->a1, &b->a1); }

#endif
