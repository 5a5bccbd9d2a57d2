"""CPU tests of the oracle itself: reference constants (the only data the reference pins for this path),
Python oracle self-consistency, and the C restatement against the Python oracle / committed vectors."""
import numpy as np
import pytest


# ---- reference KATs (config/statetransition_vkey.sol:42-114, circuits/voteverifier/dummy.go:15-47) ------------
def test_moduli_match_reference(opy, refconst):
    assert opy.P == int(refconst["P"]) and opy.R == int(refconst["R"])


def test_fp_constants_match_reference(opy, refconst):
    P = opy.P
    assert int(refconst["FRACTION_1_2_FP"]) == pow(2, -1, P)
    assert int(refconst["FRACTION_27_82_FP"]) == 27 * pow(82, -1, P) % P
    assert int(refconst["FRACTION_3_82_FP"]) == 3 * pow(82, -1, P) % P
    assert int(refconst["EXP_INVERSE_FP"]) == P - 2
    assert int(refconst["EXP_SQRT_FP"]) == (P + 1) // 4
    # 27/82 = real part of 3/(9+u): pins the twist constant convention Fp2 = Fp[u]/(u^2+1), A0 + A1 u
    assert opy.B_G2 == (int(refconst["FRACTION_27_82_FP"]), (-int(refconst["FRACTION_3_82_FP"])) % P)


def test_reference_g1_points_on_curve(opy, orc, refconst, conv):
    for name, (x, y) in refconst["g1_points"].items():
        pt = (int(x), int(y))
        assert opy.g1_on_curve(pt), name
        assert orc.g1_on_curve(conv.g1([pt])[0]), name
    a = refconst["snarkjs_proof"]
    assert opy.g1_on_curve((int(a["pi_a"][0]), int(a["pi_a"][1])))
    assert opy.g1_on_curve((int(a["pi_c"][0]), int(a["pi_c"][1])))


def test_reference_g2_points_on_twist(opy, orc, refconst, conv):
    for name, ((x0, x1), (y0, y1)) in refconst["g2_points"].items():
        pt = ((int(x0), int(x1)), (int(y0), int(y1)))
        assert opy.g2_on_curve(pt), name
        assert orc.g2_on_curve(conv.g2([[x0, x1, y0, y1]])[0].astype(np.uint64) if False else conv.g2([[int(x0), int(x1), int(y0), int(y1)]])[0]), name
        assert opy.g2_mul(pt, opy.R) is None, name + " not in the r-torsion"
    b = refconst["snarkjs_proof"]["pi_b"]
    assert opy.g2_on_curve(((int(b[0][0]), int(b[0][1])), (int(b[1][0]), int(b[1][1]))))


def test_generators(opy, orc):
    assert opy.g1_on_curve(opy.G1_GEN) and opy.g2_on_curve(opy.G2_GEN)
    assert opy.g1_mul(opy.G1_GEN, opy.R) is None and opy.g2_mul(opy.G2_GEN, opy.R) is None
    assert list(orc.g1_generator()) == opy.g1_to_limbs(opy.G1_GEN)
    assert list(orc.g2_generator()) == opy.g2_to_limbs(opy.G2_GEN)


def test_montgomery_constants(opy):
    assert opy.mont_constants(opy.P)["inv64"] == 0x87D20782E4866389
    assert opy.mont_constants(opy.R)["inv64"] == 0xC2E1F593EFFFFFFF
    assert opy.mont_constants(opy.P)["R_mod"] & 0xFFFFFFF == 0x58F0D9D & 0xFFFFFFF


# ---- Python oracle conventions ---------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [1, 2, 4, 5])
def test_py_fft_conventions_vs_naive_dft(opy, log_n):
    o = opy
    d = o.Domain(1 << log_n); n = d.n
    a = [o.fr_at(7, i) for i in range(n)]
    ref = o.naive_dft(a, d.gen)
    x = list(a); d.fft(x, o.DIF)
    assert [x[o.bitrev(i, log_n)] for i in range(n)] == ref          # DIF: natural in, bit-reversed out
    xb = [a[o.bitrev(i, log_n)] for i in range(n)]; d.fft(xb, o.DIT)
    assert xb == ref                                                  # DIT: bit-reversed in, natural out
    x = list(a); d.fft(x, o.DIF); d.fft_inverse(x, o.DIT); assert x == a
    refc = o.naive_dft([a[i] * pow(5, i, o.R) % o.R for i in range(n)], d.gen)
    x = list(a); d.fft(x, o.DIF, coset=True)
    assert [x[o.bitrev(i, log_n)] for i in range(n)] == refc
    d.fft_inverse(x, o.DIT, coset=True); assert x == a
    x = [a[o.bitrev(i, log_n)] for i in range(n)]; d.fft(x, o.DIT, coset=True); assert x == refc
    d.fft_inverse(x, o.DIF, coset=True); assert [x[o.bitrev(i, log_n)] for i in range(n)] == a


def test_py_compute_h_is_quotient(opy):
    """h from computeH (bit-reversed coefficients) satisfies a(x) b(x) - c(x) = h(x) (x^n - 1) at a random point."""
    o = opy
    cs = o.SyntheticR1CS(13, 3, 11)
    a, b, c = cs.abc()
    d = o.Domain(13); n = d.n
    h = o.compute_h(a, b, c, d)
    hc = [h[o.bitrev(i, d.log_n)] for i in range(n)]
    def interp(vals):
        v = list(vals) + [0] * (n - len(vals)); d.fft_inverse(v, o.DIF); return [v[o.bitrev(i, d.log_n)] for i in range(n)]
    ev = lambda cf, x: sum(cf_i * pow(x, i, o.R) for i, cf_i in enumerate(cf)) % o.R
    x = 0x1234567
    A, B, C = interp(a), interp(b), interp(c)
    assert (ev(A, x) * ev(B, x) - ev(C, x)) % o.R == ev(hc, x) * (pow(x, n, o.R) - 1) % o.R
    assert hc[n - 1] == 0          # deg h <= n-2: why gnark's MSM uses h[:n-1]


def test_py_groth16_trapdoor_identity(opy):
    o = opy
    cs = o.SyntheticR1CS(13, 3, 11)
    pk, trap = o.groth16_setup(cs, 3)
    pr = o.groth16_prove(cs, pk, 1234567, 7654321)
    ex = o.groth16_expected_from_trapdoor(cs, trap, pk["domain"], 1234567, 7654321)
    assert pr["Ar"] == ex["Ar"] and pr["Bs"] == ex["Bs"] and pr["Krs"] == ex["Krs"]


# ---- C restatement vs Python oracle / golden vectors -----------------------------------------------------------
def test_c_field_ops(opy, orc):
    o = opy
    L = lambda x: np.array(o.limbs(x), dtype=np.uint64)
    V = lambda a: o.from_limbs(a)
    rng = o.SplitMix64(99)
    for it in range(300):
        a = rng.next_fr() * 3 % o.P; b = rng.next_fr() * 5 % o.P
        if it == 0: a = 0
        if it == 1: a = b = o.P - 1
        am, bm = o.to_mont(a, o.P), o.to_mont(b, o.P)
        assert V(orc.fp_mul(L(am), L(bm))) == o.to_mont(a * b % o.P, o.P)
        assert V(orc.fp_add(L(am), L(bm))) == o.to_mont((a + b) % o.P, o.P)
        assert V(orc.fp_sub(L(am), L(bm))) == o.to_mont((a - b) % o.P, o.P)
        a %= o.R; b %= o.R; am, bm = o.to_mont(a, o.R), o.to_mont(b, o.R)
        assert V(orc.fr_mul(L(am), L(bm))) == o.to_mont(a * b % o.R, o.R)
        assert V(orc.fr_from_mont(L(am))) == a
    assert V(orc.fp_inv(L(o.to_mont(12345, o.P)))) == o.to_mont(pow(12345, -1, o.P), o.P)
    assert V(orc.fr_inv(L(o.to_mont(12345, o.R)))) == o.to_mont(pow(12345, -1, o.R), o.R)


def test_c_synthetic_inputs_match_python(opy, orc):
    o = opy
    pts = orc.g1_synthetic_points(0xDA71C1, 200)
    pp = o.synthetic_g1_points(0xDA71C1, 200)
    assert [list(r) for r in pts] == [o.g1_to_limbs(p) for p in pp]
    pts2 = orc.g2_synthetic_points(5, 30)
    assert [list(r) for r in pts2] == [o.g2_to_limbs(p) for p in o.synthetic_g2_points(5, 30)]
    for kind, f in ((0, o.fr_at), (1, o.witness_like_scalar)):
        sc = orc.scalars(1, kind, 300)
        assert [o.from_mont(o.from_limbs(r), o.R) for r in sc] == [f(1, i) for i in range(300)]
    assert all(orc.g1_on_curve(p) for p in pts[:20]) and all(orc.g2_on_curve(p) for p in pts2[:5])


def test_c_msm_g1_golden(orc, vectors, conv):
    for case in vectors["msm_g1"]:
        pts = conv.g1(case["points"]); sc = conv.fr(case["scalars"])
        want = conv.g1([case["result"]])[0]
        assert (orc.msm_g1(pts, sc) == want).all(), case["name"]
        assert (orc.msm_g1(pts, sc, nthreads=1) == want).all(), case["name"]


def test_c_msm_g2_golden(orc, vectors, conv):
    for case in vectors["msm_g2"]:
        pts = conv.g2(case["points"]); sc = conv.fr(case["scalars"])
        assert (orc.msm_g2(pts, sc) == conv.g2([case["result"]])[0]).all(), case["name"]


def test_c_msm_errors_and_empty(orc):
    pts = orc.g1_synthetic_points(1, 4); sc = orc.scalars(1, 0, 3)
    with pytest.raises(ValueError):
        orc.msm_g1(pts, sc)                       # MultiExp: len(points) != len(scalars)
    assert (orc.msm_g1(pts[:0], sc[:0]) == 0).all()      # empty -> infinity (0,0)


@pytest.mark.parametrize("n", [1000, 1 << 13])
def test_c_msm_closed_form(opy, orc, n):
    """sum s_i (P0 + i D) = (sum s_i) P0 + (sum i s_i) D : independent of the Pippenger code path."""
    o = opy
    pts = orc.g1_synthetic_points(0xDA71C1, n)
    for kind in (0, 1):
        sc = orc.scalars(3, kind, n)
        vals = [o.from_limbs(r) for r in orc.fr_vec_from_mont(sc)]
        s0 = sum(vals) % o.R; s1 = sum(i * v for i, v in enumerate(vals)) % o.R
        p0 = o.g1_from_limbs([int(v) for v in pts[0]]); p1 = o.g1_from_limbs([int(v) for v in pts[1]])
        d = o.g1_add(p1, o.g1_neg(p0))
        want = o.g1_add(o.g1_mul(p0, s0), o.g1_mul(d, s1))
        assert list(orc.msm_g1(pts, sc)) == o.g1_to_limbs(want)


def test_c_ntt_golden(orc, vectors, conv):
    for case in vectors["ntt"]:
        x = conv.fr(case["input"])
        y = orc.ntt(x, bool(case["inverse"]), case["decimation"], bool(case["coset"]))
        assert (y == conv.fr(case["output"])).all(), case


def test_c_ntt_vs_python_medium(opy, orc, conv):
    o = opy
    n = 1 << 9; d = o.Domain(n)
    x = [o.fr_at(2, i) for i in range(n)]; xm = conv.fr(x)
    for inv in (0, 1):
        for dec in (0, 1):
            for cos in (0, 1):
                y = list(x); (d.fft_inverse if inv else d.fft)(y, dec, bool(cos))
                assert conv.fr_back(orc.ntt(xm, bool(inv), dec, bool(cos))) == y


def test_c_compute_h_golden(orc, vectors, conv):
    for case in vectors["compute_h"]:
        h = orc.compute_h(conv.fr(case["a"]), conv.fr(case["b"]), conv.fr(case["c"]), case["log_n"])
        assert (h == conv.fr(case["h"])).all()


def test_c_ntt_roundtrip_large(orc):
    x = orc.scalars(2, 0, 1 << 16)
    y = orc.ntt(orc.ntt(x, False, 0, False), True, 1, False)
    assert (x == y).all()
