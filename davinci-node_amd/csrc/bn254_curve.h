// BN254 G1 / G2 group law for the MSM kernels: extended Jacobian (X, Y, ZZ, ZZZ) buckets + affine points.
//
// Stands in for gnark-crypto's g1JacExtended / g2JacExtended bucket arithmetic inside G1Jac.MultiExp /
// G2Jac.MultiExp (reached from /root/reference/prover/prover_cpu.go:37).
// Formulas: EFD madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1 (a = 0).  Affine infinity is (0,0);
// a bucket at infinity has ZZ == literal 0.
#pragma once
#include "bn254_field.h"

namespace dvn {

template <class F> struct Affine { F x, y; };
template <class F> struct Xyzz { F x, y, zz, zzz; };

template <class F> struct Curve {
    typedef Ops<F> O;
    typedef Affine<F> A;
    typedef Xyzz<F> X;

    static DVN_HD X infinity() { return X{O::one(), O::one(), O::zero(), O::zero()}; }
    static DVN_HD bool is_inf(const X& p) { return O::is_literal_zero(p.zz); }
    static DVN_HD bool is_inf(const A& p) { return O::is_literal_zero(p.x) && O::is_literal_zero(p.y); }
    static DVN_HD X from_affine(const A& p) { return is_inf(p) ? infinity() : X{p.x, p.y, O::one(), O::one()}; }
    static DVN_HD A neg(const A& p) { return A{p.x, O::neg(p.y)}; }
    static DVN_HD X neg(const X& p) { return X{p.x, O::neg(p.y), p.zz, p.zzz}; }

    // 2*q for affine q != infinity (mdbl-2008-s-1)
    static DVN_HD X dbl_affine(const A& q) {
        F U = O::dbl(q.y), V = O::sqr(U), W = O::mul(U, V), S = O::mul(q.x, V);
        F xx = O::sqr(q.x), M = O::add(O::dbl(xx), xx);
        F X3 = O::sub(O::sub(O::sqr(M), S), S);
        F Y3 = O::sub(O::mul(M, O::sub(S, X3)), O::mul(W, q.y));
        return X{X3, Y3, V, W};
    }
    // 2*p (dbl-2008-s-1)
    static DVN_HD X dbl_impl(const X& p) {
        if (is_inf(p)) return p;
        F U = O::dbl(p.y), V = O::sqr(U), W = O::mul(U, V), S = O::mul(p.x, V);
        F xx = O::sqr(p.x), M = O::add(O::dbl(xx), xx);
        F X3 = O::sub(O::sub(O::sqr(M), S), S);
        F Y3 = O::sub(O::mul(M, O::sub(S, X3)), O::mul(W, p.y));
        return X{X3, Y3, O::mul(V, p.zz), O::mul(W, p.zzz)};
    }
    // p + q, q affine (madd-2008-s); q may be infinity
    static DVN_HD X add_mixed_impl(const X& p, const A& q) {
        if (is_inf(q)) return p;
        if (is_inf(p)) return X{q.x, q.y, O::one(), O::one()};
        F U2 = O::mul(q.x, p.zz), S2 = O::mul(q.y, p.zzz);
        F P = O::sub(U2, p.x), R = O::sub(S2, p.y);
        if (O::is_zero(P)) {
            if (O::is_zero(R)) return dbl_affine(q);
            return infinity();
        }
        F PP = O::sqr(P), PPP = O::mul(P, PP), Q = O::mul(p.x, PP);
        F X3 = O::sub(O::sub(O::sub(O::sqr(R), PPP), Q), Q);
        F Y3 = O::sub(O::mul(R, O::sub(Q, X3)), O::mul(p.y, PPP));
        return X{X3, Y3, O::mul(p.zz, PP), O::mul(p.zzz, PPP)};
    }
    // p + q (add-2008-s)
    static DVN_HD X add_impl(const X& p, const X& q) {
        if (is_inf(q)) return p;
        if (is_inf(p)) return q;
        F U1 = O::mul(p.x, q.zz), U2 = O::mul(q.x, p.zz);
        F S1 = O::mul(p.y, q.zzz), S2 = O::mul(q.y, p.zzz);
        F P = O::sub(U2, U1), R = O::sub(S2, S1);
        if (O::is_zero(P)) {
            if (O::is_zero(R)) return dbl_impl(p);
            return infinity();
        }
        F PP = O::sqr(P), PPP = O::mul(P, PP), Q = O::mul(U1, PP);
        F X3 = O::sub(O::sub(O::sub(O::sqr(R), PPP), Q), Q);
        F Y3 = O::sub(O::mul(R, O::sub(Q, X3)), O::mul(S1, PPP));
        return X{X3, Y3, O::mul(O::mul(p.zz, q.zz), PP), O::mul(O::mul(p.zzz, q.zzz), PPP)};
    }
    // Accumulator form for the MSM hot loop (G1): coordinates are LOOSE -- x < 8p, y < 2p, zz, zzz < 2p, limbs
    // normalised -- so the additive steps are single carry passes and Y3 shares one Montgomery reduction between its
    // two products.  Bounds (units of p): P = U2 + 8p - X1 < 9.1, R = S2 + 2p - Y1 < 3.1, PP < 1.5, PPP, Q, RR < 1.1,
    // X3 = RR + 2p - PPP + 4p - 2Q < 7.1, QX = Q + 8p - X3 < 9.1, Y3 = (R QX + (2p - Y1) PPP)/R' < 1.2.
    static DVN_HD X acc_add_mixed(const X& p, const A& q) { return acc_add_mixed_signed(p, q, false); }
    // p + q, or p - q when `negate`: the sign of a sorted MSM entry is folded into the one subtraction that reads q.y
    // (R = +-S2 - Y1) instead of negating the point first
    static DVN_HD X acc_add_mixed_signed(const X& p, const A& q, bool negate) {
        if constexpr (!O::LAZY) {
            return add_mixed(p, negate ? neg(q) : q);
        } else {
            typedef typename O::Cfg C;
            if (is_inf(q)) return p;
            if (is_inf(p)) return X{q.x, negate ? O::neg(q.y) : q.y, O::one(), O::one()};
            const F U2 = O::mul(q.x, p.zz), S2 = O::mul(q.y, p.zzz);
            const F P = fe_sub_lazy<C, 8>(U2, p.x), R = fe_sub_lazy_cneg<C>(S2, p.y, negate);
            if (fe_is_zero_loose<C>(P)) {
                if (fe_is_zero_loose<C>(R)) return dbl_affine(negate ? neg(q) : q);
                return infinity();
            }
            const F PP = O::sqr(P), PPP = O::mul(P, PP), Q = O::mul(p.x, PP);
            // products are below 1.7p here (P < 10p, R < 4p, X1 < 8p): PPP + 2Q < 6p, X3 < R^2 + 6p < 8p
            const F X3 = fe_sub2_lazy6<C>(O::sqr(R), PPP, Q);
            F D;                                                  // Q - X3 + K p, the second operand of R in Y3
            if constexpr (fe_mul_add_raw_fits<C>()) D = fe_sub_raw<C, 16, LB>(Q, X3);     // raw limbs: no carry pass
            else D = fe_sub_lazy<C, 8>(Q, X3);
            F NY;                                                 // K p - Y1 (Y1 < 2p), the first operand of PPP in Y3
            if constexpr (fe_mul_add_raw2_fits<C>()) NY = fe_neg_raw<C, 4, LB>(p.y);
            else NY = fe_neg_lazy<C, 2>(p.y);
            const F Y3 = fe_mul_add<C>(D, R, NY, PPP);
            return X{X3, Y3, O::mul(p.zz, PP), O::mul(p.zzz, PPP)};
        }
    }
    // loose accumulator -> weak form (before it is stored or handed to the regular group law)
    static DVN_HD X acc_normalize(const X& p) {
        if constexpr (!O::LAZY) return p;
        else {
            if (is_inf(p)) return p;
            typedef typename O::Cfg C;
            return X{fe_reduce_loose<C>(p.x), p.y, p.zz, p.zzz};
        }
    }

    // Group operations.  For G2 (Fp2: ~3x the code of G1 per operation) they are real calls: fully inlined,
    // the G2 kernels exceed 10^5 instructions and need long-branch relaxation; outlined they stay small.
    static DVN_NOINLINE X dbl_out(const X& p) { return dbl_impl(p); }
    static DVN_NOINLINE X add_out(const X& p, const X& q) { return add_impl(p, q); }
    static DVN_NOINLINE X add_mixed_out(const X& p, const A& q) { return add_mixed_impl(p, q); }
    static DVN_HD X dbl(const X& p) { if constexpr (O::OUTLINE) return dbl_out(p); else return dbl_impl(p); }
    static DVN_HD X add(const X& p, const X& q) { if constexpr (O::OUTLINE) return add_out(p, q); else return add_impl(p, q); }
    static DVN_HD X add_mixed(const X& p, const A& q) { if constexpr (O::OUTLINE) return add_mixed_out(p, q); else return add_mixed_impl(p, q); }
    // [k]p for a small non-negative integer k (MSB-first double-and-add)
    static DVN_HD X mul_small(const X& p, u32 k) {
        X acc = infinity();
        for (int i = 31; i >= 0; --i) {
            acc = dbl(acc);
            if ((k >> i) & 1) acc = add(acc, p);
        }
        return acc;
    }
    static DVN_HD A to_affine(const X& p) {
        if (is_inf(p)) return A{O::zero(), O::zero()};
        F zi = O::inv(p.zzz);                       // 1/ZZZ
        F iz2 = O::sqr(O::mul(zi, p.zz));           // (ZZ/ZZZ)^2 = 1/ZZ
        return A{O::mul(p.x, iz2), O::mul(p.y, zi)};
    }

    // ---- memory images ----------------------------------------------------------------------------------
    // Device-resident points/buckets are stored in *internal* Montgomery form (R' = 2^261) packed as
    // 8 x u32 per Fp coordinate, so a load is pure bit slicing.  AW / XW = u32 words per affine / xyzz.
    static constexpr int AW = 2 * O::WORDS;
    static constexpr int XW = 4 * O::WORDS;
    static DVN_HD A load_affine_internal(const u32* w) { return A{O::load_internal(w), O::load_internal(w + O::WORDS)}; }
    static DVN_HD void store_affine_internal(const A& p, u32* w) { O::store_internal(p.x, w); O::store_internal(p.y, w + O::WORDS); }
    static DVN_HD X load_xyzz_internal(const u32* w) {
        return X{O::load_internal(w), O::load_internal(w + O::WORDS), O::load_internal(w + 2 * O::WORDS), O::load_internal(w + 3 * O::WORDS)};
    }
    static DVN_HD void store_xyzz_internal(const X& p, u32* w) {
        if (is_inf(p)) {          // canonical image of infinity: all-zero ZZ/ZZZ (X, Y arbitrary -> 0)
            for (int i = 0; i < XW; ++i) w[i] = 0;
            return;
        }
        O::store_internal(p.x, w); O::store_internal(p.y, w + O::WORDS);
        O::store_internal(p.zz, w + 2 * O::WORDS); O::store_internal(p.zzz, w + 3 * O::WORDS);
    }
    // gnark affine image (Montgomery R = 2^256) <-> internal
    static DVN_HD A affine_from_ext(const u32* w) {
        bool z = true;
        for (int i = 0; i < AW; ++i) z = z && (w[i] == 0);
        if (z) return A{O::zero(), O::zero()};
        return A{O::from_ext(w), O::from_ext(w + O::WORDS)};
    }
    static DVN_HD void affine_to_ext(const A& p, u32* w) {
        if (is_inf(p)) { for (int i = 0; i < AW; ++i) w[i] = 0; return; }
        O::to_ext(p.x, w); O::to_ext(p.y, w + O::WORDS);
    }
};

typedef Curve<Fp> G1;
typedef Curve<Fp2> G2;
// BLS12-377 (voteverifier curve): same a = 0 group law over 14-limb base-field elements
typedef Curve<Fp377> G1_377;
typedef Curve<Fp2_377> G2_377;
typedef Curve<Fp381> G1_381;
typedef Curve<Fp2_381> G2_381;
// BW6-761 (aggregator curve): G1 and G2 are both curves over the 761-bit base field (27 limbs)
typedef Curve<Fp761> G_761;

}  // namespace dvn
