/* Curve-group template for the CPU oracle (TEST INFRASTRUCTURE).  Included twice by bn254_cpu.c:
 *   G1 over Fp   (FE = fe,  F(op) = fp_op,  ECN(x) = g1_x)
 *   G2 over Fp2  (FE = fe2, F(op) = fp2_op, ECN(x) = g2_x)
 * y^2 = x^3 + b, a = 0.  Affine infinity = (0,0) as in gnark-crypto's G1Affine/G2Affine.
 * Buckets use extended Jacobian (X, Y, ZZ, ZZZ) like gnark-crypto's g1JacExtended (ecc/bn254/g1.go, EFD
 * madd-2008-s / add-2008-s / dbl-2008-s-1), and the MSM follows ecc/bn254/multiexp.go's structure:
 * signed c-bit digits, one task per window, running-sum bucket reduction, MSB-first window combine.
 * The batch-affine bucket variant gnark uses for c >= 10 computes the same group element and is not restated.
 */

typedef struct { FE x, y; } ECN(aff);
typedef struct { FE x, y, zz, zzz; } ECN(xyzz);

static inline int ECN(aff_is_inf)(const ECN(aff)* p) { return F(is_zero)(&p->x) && F(is_zero)(&p->y); }
static inline void ECN(xyzz_set_inf)(ECN(xyzz)* p) { F(set_one)(&p->x); F(set_one)(&p->y); F(set_zero)(&p->zz); F(set_zero)(&p->zzz); }
static inline int ECN(xyzz_is_inf)(const ECN(xyzz)* p) { return F(is_zero)(&p->zz); }

static inline void ECN(xyzz_from_aff)(ECN(xyzz)* r, const ECN(aff)* p) {
    if (ECN(aff_is_inf)(p)) { ECN(xyzz_set_inf)(r); return; }
    r->x = p->x; r->y = p->y; F(set_one)(&r->zz); F(set_one)(&r->zzz);
}

/* r = 2*p for affine p (mdbl-2008-s-1) */
static inline void ECN(xyzz_dbl_aff)(ECN(xyzz)* r, const ECN(aff)* p) {
    FE U, V, W, S, M, t;
    F(dbl)(&U, &p->y); F(sqr)(&V, &U); F(mul)(&W, &U, &V); F(mul)(&S, &p->x, &V);
    F(sqr)(&t, &p->x); F(dbl)(&M, &t); F(add)(&M, &M, &t);
    F(sqr)(&r->x, &M); F(sub)(&r->x, &r->x, &S); F(sub)(&r->x, &r->x, &S);
    F(sub)(&t, &S, &r->x); F(mul)(&t, &M, &t); F(mul)(&U, &W, &p->y); F(sub)(&r->y, &t, &U);
    r->zz = V; r->zzz = W;
}

/* p = 2*p (dbl-2008-s-1) */
static inline void ECN(xyzz_dbl)(ECN(xyzz)* p) {
    if (ECN(xyzz_is_inf)(p)) return;
    FE U, V, W, S, M, t, X3;
    F(dbl)(&U, &p->y); F(sqr)(&V, &U); F(mul)(&W, &U, &V); F(mul)(&S, &p->x, &V);
    F(sqr)(&t, &p->x); F(dbl)(&M, &t); F(add)(&M, &M, &t);
    F(sqr)(&X3, &M); F(sub)(&X3, &X3, &S); F(sub)(&X3, &X3, &S);
    F(sub)(&t, &S, &X3); F(mul)(&t, &M, &t); F(mul)(&U, &W, &p->y); F(sub)(&p->y, &t, &U);
    p->x = X3; F(mul)(&p->zz, &V, &p->zz); F(mul)(&p->zzz, &W, &p->zzz);
}

/* p += (neg ? -q : q), q affine (madd-2008-s) */
static inline void ECN(xyzz_add_aff)(ECN(xyzz)* p, const ECN(aff)* q, int neg) {
    if (ECN(aff_is_inf)(q)) return;
    ECN(aff) qq = *q;
    if (neg) F(neg)(&qq.y, &q->y);
    if (ECN(xyzz_is_inf)(p)) { p->x = qq.x; p->y = qq.y; F(set_one)(&p->zz); F(set_one)(&p->zzz); return; }
    FE U2, S2, Pq, Rq, PP, PPP, Q, t;
    F(mul)(&U2, &qq.x, &p->zz); F(mul)(&S2, &qq.y, &p->zzz);
    F(sub)(&Pq, &U2, &p->x); F(sub)(&Rq, &S2, &p->y);
    if (F(is_zero)(&Pq)) {
        if (F(is_zero)(&Rq)) { ECN(xyzz_dbl_aff)(p, &qq); return; }
        ECN(xyzz_set_inf)(p); return;
    }
    F(sqr)(&PP, &Pq); F(mul)(&PPP, &Pq, &PP); F(mul)(&Q, &p->x, &PP);
    F(sqr)(&t, &Rq); F(sub)(&t, &t, &PPP); F(sub)(&t, &t, &Q); F(sub)(&t, &t, &Q);   /* X3 */
    F(sub)(&Q, &Q, &t); F(mul)(&Q, &Rq, &Q); F(mul)(&U2, &p->y, &PPP); F(sub)(&p->y, &Q, &U2);
    p->x = t; F(mul)(&p->zz, &p->zz, &PP); F(mul)(&p->zzz, &p->zzz, &PPP);
}

/* p += q (add-2008-s) */
static inline void ECN(xyzz_add)(ECN(xyzz)* p, const ECN(xyzz)* q) {
    if (ECN(xyzz_is_inf)(q)) return;
    if (ECN(xyzz_is_inf)(p)) { *p = *q; return; }
    FE U1, U2, S1, S2, Pq, Rq, PP, PPP, Q, t;
    F(mul)(&U1, &p->x, &q->zz); F(mul)(&U2, &q->x, &p->zz);
    F(mul)(&S1, &p->y, &q->zzz); F(mul)(&S2, &q->y, &p->zzz);
    F(sub)(&Pq, &U2, &U1); F(sub)(&Rq, &S2, &S1);
    if (F(is_zero)(&Pq)) {
        if (F(is_zero)(&Rq)) { ECN(xyzz_dbl)(p); return; }
        ECN(xyzz_set_inf)(p); return;
    }
    F(sqr)(&PP, &Pq); F(mul)(&PPP, &Pq, &PP); F(mul)(&Q, &U1, &PP);
    F(sqr)(&t, &Rq); F(sub)(&t, &t, &PPP); F(sub)(&t, &t, &Q); F(sub)(&t, &t, &Q);
    F(sub)(&Q, &Q, &t); F(mul)(&Q, &Rq, &Q); F(mul)(&U2, &S1, &PPP); F(sub)(&p->y, &Q, &U2);
    p->x = t;
    F(mul)(&p->zz, &p->zz, &q->zz); F(mul)(&p->zz, &p->zz, &PP);
    F(mul)(&p->zzz, &p->zzz, &q->zzz); F(mul)(&p->zzz, &p->zzz, &PPP);
}

static inline void ECN(xyzz_to_aff)(ECN(aff)* r, const ECN(xyzz)* p) {
    if (ECN(xyzz_is_inf)(p)) { F(set_zero)(&r->x); F(set_zero)(&r->y); return; }
    FE zi, zi2;                       /* x = X/ZZ, y = Y/ZZZ; 1/ZZ = ZZ^2... use 1/ZZZ and ZZ: (1/ZZZ)*... */
    F(inv)(&zi, &p->zzz);             /* zi = 1/ZZZ */
    F(mul)(&r->y, &p->y, &zi);
    F(mul)(&zi2, &zi, &p->zz);        /* ZZ/ZZZ = 1/Z */
    F(sqr)(&zi2, &zi2);               /* 1/Z^2 = 1/ZZ */
    F(mul)(&r->x, &p->x, &zi2);
}

/* r = [k]p, k = 256-bit little-endian canonical integer; MSB-first double-and-add */
static void ECN(scalar_mul)(ECN(xyzz)* r, const ECN(aff)* p, const u64 k[4]) {
    ECN(xyzz) acc; ECN(xyzz_set_inf)(&acc);
    for (int i = 255; i >= 0; --i) {
        ECN(xyzz_dbl)(&acc);
        if ((k[i >> 6] >> (i & 63)) & 1) ECN(xyzz_add_aff)(&acc, p, 0);
    }
    *r = acc;
}

/* out[i] affine of in[i] with one shared inversion (Montgomery's trick); infinity -> (0,0) */
static void ECN(batch_to_aff)(ECN(aff)* out, const ECN(xyzz)* in, size_t n) {
    FE* pre = (FE*)malloc(sizeof(FE) * (n + 1));
    FE acc; F(set_one)(&acc);
    for (size_t i = 0; i < n; ++i) {
        pre[i] = acc;
        if (!ECN(xyzz_is_inf)(&in[i])) F(mul)(&acc, &acc, &in[i].zzz);
    }
    FE inv; F(inv)(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (ECN(xyzz_is_inf)(&in[i])) { F(set_zero)(&out[i].x); F(set_zero)(&out[i].y); continue; }
        FE zi, t; F(mul)(&zi, &inv, &pre[i]);            /* 1/ZZZ_i */
        F(mul)(&inv, &inv, &in[i].zzz);
        F(mul)(&out[i].y, &in[i].y, &zi);
        F(mul)(&t, &zi, &in[i].zz); F(sqr)(&t, &t);
        F(mul)(&out[i].x, &in[i].x, &t);
    }
    free(pre);
}

/* Pippenger.  scalars are CANONICAL (non-Montgomery) 256-bit little-endian, already < r. */
static int ECN(pippenger_c)(size_t n) {
    /* gnark-crypto's cost model: minimise (bits+1)/c * (n + 2^c) over its implemented window sizes */
    static const int cs[] = {4, 5, 8, 10, 11, 12, 13, 14, 15, 16};
    int best = 4; double bc = 1e300;
    for (unsigned i = 0; i < sizeof(cs) / sizeof(cs[0]); ++i) {
        int c = cs[i];
        double cost = (double)((255 + c - 1) / c) * ((double)n + (double)(1ull << c));
        if (cost < bc) { bc = cost; best = c; }
    }
    return best;
}

/* c_override > 0 forces the window size; wsum_out (optional) receives the per-window sums, 254/c + 1 of them */
static void ECN(msm_ex)(ECN(xyzz)* result, const ECN(aff)* points, const u64* scalars /* n x 4 canonical */, size_t n, int nthreads,
                        int c_override, ECN(xyzz)* wsum_out) {
    ECN(xyzz_set_inf)(result);
    if (n == 0) return;
    const int c = c_override > 0 ? c_override : ECN(pippenger_c)(n);
    const int nwin = 254 / c + 1;                 /* +1 absorbs the final signed-digit carry */
    const size_t nb = (size_t)1 << (c - 1);
    /* signed digits: digits[w*n + i] in [-2^(c-1), 2^(c-1)] */
    int32_t* digits = (int32_t*)malloc(sizeof(int32_t) * n * (size_t)nwin);
    #pragma omp parallel for num_threads(nthreads) schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const u64* s = scalars + 4 * i;
        int carry = 0;
        for (int w = 0; w < nwin; ++w) {
            int bit = w * c;
            int64_t d = carry;
            if (bit < 256) {
                u64 v = s[bit >> 6] >> (bit & 63);
                if ((bit & 63) + c > 64 && (bit >> 6) + 1 < 4) v |= s[(bit >> 6) + 1] << (64 - (bit & 63));
                d += (int64_t)(v & (((u64)1 << c) - 1));
            }
            carry = 0;
            if (d > (int64_t)nb) { d -= (int64_t)1 << c; carry = 1; }
            digits[(size_t)w * n + i] = (int32_t)d;
        }
    }
    /* one task per (window, point chunk) so that more threads than windows are used -- gnark likewise splits
     * heavy chunks; every task owns its buckets and reduces them, the per-window sums of the chunks are added */
    ECN(xyzz)* wsum = (ECN(xyzz)*)malloc(sizeof(ECN(xyzz)) * nwin);
    int nchunk = (nthreads + nwin - 1) / nwin;
    if (nchunk < 1) nchunk = 1;
    while (nchunk > 1 && n / (size_t)nchunk < 4 * nb) --nchunk;      /* keep chunks much larger than the bucket count */
    const int ntask = nwin * nchunk;
    ECN(xyzz)* tsum = (ECN(xyzz)*)malloc(sizeof(ECN(xyzz)) * ntask);
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int task = 0; task < ntask; ++task) {
        const int w = task / nchunk, ch = task % nchunk;
        const size_t lo = n * (size_t)ch / nchunk, hi = n * (size_t)(ch + 1) / nchunk;
        ECN(xyzz)* buckets = (ECN(xyzz)*)malloc(sizeof(ECN(xyzz)) * nb);
        for (size_t b = 0; b < nb; ++b) ECN(xyzz_set_inf)(&buckets[b]);
        const int32_t* dg = digits + (size_t)w * n;
        for (size_t i = lo; i < hi; ++i) {
            int32_t d = dg[i];
            if (d == 0) continue;
            if (d > 0) ECN(xyzz_add_aff)(&buckets[d - 1], &points[i], 0);
            else ECN(xyzz_add_aff)(&buckets[-d - 1], &points[i], 1);
        }
        ECN(xyzz) run, tot; ECN(xyzz_set_inf)(&run); ECN(xyzz_set_inf)(&tot);
        for (size_t b = nb; b-- > 0;) {
            ECN(xyzz_add)(&run, &buckets[b]);
            ECN(xyzz_add)(&tot, &run);
        }
        tsum[task] = tot;
        free(buckets);
    }
    for (int w = 0; w < nwin; ++w) {
        wsum[w] = tsum[w * nchunk];
        for (int ch = 1; ch < nchunk; ++ch) ECN(xyzz_add)(&wsum[w], &tsum[w * nchunk + ch]);
    }
    free(tsum);
    ECN(xyzz) acc; ECN(xyzz_set_inf)(&acc);
    for (int w = nwin - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) ECN(xyzz_dbl)(&acc);
        ECN(xyzz_add)(&acc, &wsum[w]);
    }
    *result = acc;
    if (wsum_out) memcpy(wsum_out, wsum, sizeof(ECN(xyzz)) * nwin);
    free(wsum); free(digits);
}
static void ECN(msm)(ECN(xyzz)* result, const ECN(aff)* points, const u64* scalars, size_t n, int nthreads) {
    ECN(msm_ex)(result, points, scalars, n, nthreads, 0, NULL);
}

/* out[i] = base + i*step (affine), i in [0, n): chunked mixed adds + batch normalisation */
static void ECN(arith_progression)(ECN(aff)* out, const ECN(aff)* base, const ECN(aff)* step, size_t n, int nthreads) {
    const size_t CH = 4096;
    size_t nch = (n + CH - 1) / CH;
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (size_t ch = 0; ch < nch; ++ch) {
        size_t lo = ch * CH, hi = lo + CH < n ? lo + CH : n;
        ECN(xyzz)* tmp = (ECN(xyzz)*)malloc(sizeof(ECN(xyzz)) * (hi - lo));
        ECN(xyzz) cur; u64 k[4] = {lo, 0, 0, 0};
        ECN(scalar_mul)(&cur, step, k);
        ECN(xyzz_add_aff)(&cur, base, 0);
        for (size_t i = lo; i < hi; ++i) { tmp[i - lo] = cur; ECN(xyzz_add_aff)(&cur, step, 0); }
        ECN(batch_to_aff)(out + lo, tmp, hi - lo);
        free(tmp);
    }
}
