"""BLS12-377 (the voteverifier curve) and BLS12-381 (blob KZG, the fourth curve of
prover_gpu.go's dispatch): the same HIP kernels instantiated on 14-limb base-field elements, checked against the
curve-generic Python oracle (oracle/curves_py.py).  Parity unpinned by the reference (no vectors); the oracle's
constants are cross-checked in test_bls12377_constants / test_bls12381_constants."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "davinci-node_amd", "csrc")


@pytest.fixture(scope="module", params=["bls12-377", "bls12-381"])
def S(request):
    from oracle import curves_py
    return curves_py.BLS12_377 if request.param == "bls12-377" else curves_py.BLS12_381


def jac_to_affine(S, j, g2=False):
    """Library output is the affine normal form with Z = 1 (or Z = 0 for infinity): strip Z."""
    n = S.fp_limbs * (2 if g2 else 1)
    z = j[2 * n:3 * n]
    if not z.any():
        return np.zeros(2 * n, dtype=np.uint64)
    return j[:2 * n].copy()


def test_bls12381_constants(opy):
    from oracle import curves_py
    S = curves_py.BLS12_381
    assert S.p.bit_length() == 381 and S.r.bit_length() == 255
    assert S.g1_on_curve(S.g1_gen) and S.g1_mul(S.g1_gen, S.r) is None
    assert (S.r - 1) % (1 << 32) == 0 and (S.r - 1) % (1 << 33) != 0
    assert S.root == 10238227357739495823651030575849232062558860180284477541189508159991286009131   # gnark-crypto fr/fft
    assert pow(S.root, 1 << 31, S.r) == S.r - 1
    assert pow(S.p - 1, (S.p - 1) // 2, S.p) == S.p - 1          # -1 is a non-residue: Fp2 = Fp[u]/(u^2+1)
    # 7 generates Fr*: r - 1 = 2^32 * 3 * 11 * 19 * (214-bit cofactor whose primes are large); the small ones:
    assert all(pow(7, (S.r - 1) // q, S.r) != 1 for q in (2, 3, 11, 19))
    q = S.g2_some_point(3)
    assert S.g2_on_curve(q) and S.g2_on_curve(S.g2_add(q, q)) and S.g2_on_curve(S.g2_mul(q, 12345))
    d = S.domain(8); a = [S.fr_at(7, i) for i in range(8)]
    ref = [sum(a[j] * pow(d.gen, i * j, S.r) for j in range(8)) % S.r for i in range(8)]
    x = list(a); d.fft(x, opy.DIF)
    assert [x[opy.bitrev(i, 3)] for i in range(8)] == ref


def test_bls12377_constants(opy):
    from oracle import curves_py
    S = curves_py.BLS12_377
    assert S.p.bit_length() == 377 and S.r.bit_length() == 253
    assert S.g1_on_curve(S.g1_gen) and S.g1_mul(S.g1_gen, S.r) is None
    assert (S.r - 1) % (1 << 47) == 0 and (S.r - 1) % (1 << 48) != 0
    assert S.root == 8065159656716812877374967518403273466521432693661810619979959746626482506078     # gnark-crypto fr/fft
    assert pow(S.root, 1 << 47, S.r) == 1 and pow(S.root, 1 << 46, S.r) == S.r - 1
    assert pow(S.p - 5, (S.p - 1) // 2, S.p) == S.p - 1          # -5 is a non-residue: Fp2 = Fp[u]/(u^2+5)
    q = S.g2_some_point(3)
    assert S.g2_on_curve(q) and S.g2_on_curve(S.g2_add(q, q)) and S.g2_on_curve(S.g2_mul(q, 12345))
    for log_n in (1, 3, 4):                                      # fft conventions vs a naive DFT
        d = S.domain(1 << log_n); n = d.n
        a = [S.fr_at(7, i) for i in range(n)]
        ref = [sum(a[j] * pow(d.gen, i * j, S.r) for j in range(n)) % S.r for i in range(n)]
        x = list(a); d.fft(x, opy.DIF)
        assert [x[opy.bitrev(i, log_n)] for i in range(n)] == ref
        d.fft_inverse(x, opy.DIT); assert x == a


def test_bls12377_host_arithmetic(S, opy):
    """The 14-limb field / Fe2 (u^2 = -5) / group law headers compiled for the host against the oracle."""
    subprocess.check_call(["make", "-C", CSRC, "-s", "libdvn_hostcheck.so"])
    lib = C.CDLL(os.path.join(CSRC, "libdvn_hostcheck.so"))
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    rng = opy.SplitMix64(5)
    def rnd(mod):
        v = 0
        for i in range(6):
            v |= rng.next() << (64 * i)
        return v % mod
    hc = lambda name: getattr(lib, ("hc377_" if S.name == "bls12-377" else "hc381_") + name)
    for mod, fn, nl, tm in ((S.p, hc("fp_op"), 6, S.fp_to_mont), (S.r, hc("fr_op"), 4, S.fr_to_mont)):
        L = lambda x: np.array(S.limbs(tm(x), nl), dtype=np.uint64)
        for it in range(600):
            a, b = rnd(mod), rnd(mod)
            if it == 0: a = 0
            if it == 1: a = b = mod - 1
            am, bm, r = L(a), L(b), np.zeros(nl, dtype=np.uint64)
            for op, w in {0: a * b % mod, 1: (a + b) % mod, 2: (a - b) % mod, 3: a * a % mod, 5: (-a) % mod, 6: 2 * a % mod}.items():
                fn(p(am), p(bm), p(r), op)
                assert opy.from_limbs(r) == tm(w), (op, hex(a), hex(b))
            if it < 6:
                fn(p(am), p(bm), p(r), 4)
                assert opy.from_limbs(r) == tm(pow(a, -1, mod) if a else 0)
    pts = S.synthetic_g1_points(9, 8); ks = [S.fr_at(3, i) for i in range(8)]; ks[3] = 0; ks[4] = 1; ks[5] = S.r - 1
    pa = np.array([S.g1_to_limbs(q) for q in pts], dtype=np.uint64); ka = np.array([S.limbs(k, 4) for k in ks], dtype=np.uint64)
    r = np.zeros(12, dtype=np.uint64); hc("g1_msm_naive")(p(pa), p(ka), C.c_size_t(8), p(r))
    assert list(r) == S.g1_to_limbs(S.g1_msm(pts, ks))
    q = S.synthetic_g2_points(4, 4); qa = np.array([S.g2_to_limbs(x) for x in q], dtype=np.uint64)
    r2 = np.zeros(24, dtype=np.uint64); hc("g2_msm_naive")(p(qa), p(ka), C.c_size_t(4), p(r2))
    assert list(r2) == S.g2_to_limbs(S.g2_msm(q, ks[:4]))
    for idx in ([0, 1, 2, 3], [5, 5], [5, 5 | (1 << 31)], [5, 5 | (1 << 31), 7], [3] * 20):
        ia = np.array(idx, dtype=np.uint32); r = np.zeros(12, dtype=np.uint64)
        hc("g1_acc_chain")(p(pa), p(ia), C.c_size_t(len(idx)), p(r))
        acc = None
        for e in idx:
            pt = pts[e & 0x7fffffff]; pt = S.g1_neg(pt) if e >> 31 else pt; acc = S.g1_add(acc, pt)
        assert list(r) == S.g1_to_limbs(acc), idx


@pytest.fixture(scope="module")
def bls(ctx, pkg, S):
    return pkg.Bls12377(ctx) if S.name == "bls12-377" else pkg.Bls12381(ctx)


@pytest.mark.gpu
def test_bls12377_msm_g1_vs_oracle(bls, ctx, S):
    pts = S.synthetic_g1_points(0xDA71C1, 300)
    pa = np.array([S.g1_to_limbs(q) for q in pts], dtype=np.uint64)
    pa[7] = 0; pts[7] = None                                                    # an infinity point
    dp = bls.upload_g1(pa)
    assert (bls.download(dp) == pa).all()
    for n in (1, 2, 33, 300):
        ks = [S.fr_at(1 + n, i) for i in range(n)]
        if n >= 33:
            ks[3], ks[4], ks[5] = 0, 1, S.r - 1
        sc = np.array([S.fr_to_limbs(k) for k in ks], dtype=np.uint64)
        got = jac_to_affine(S, bls.msm_g1(dp, sc))
        assert list(got) == S.g1_to_limbs(S.g1_msm(pts[:n], ks)), n
    # P and -P in one bucket, a repeated point, empty input, length mismatch
    edge = [pts[1], S.g1_neg(pts[1]), pts[2], pts[2], pts[2]]
    de = bls.upload_g1(np.array([S.g1_to_limbs(q) for q in edge], dtype=np.uint64))
    ks = [9, 9, 5, 5, 5]
    got = jac_to_affine(S, bls.msm_g1(de, np.array([S.fr_to_limbs(k) for k in ks], dtype=np.uint64)))
    assert list(got) == S.g1_to_limbs(S.g1_msm(edge, ks))
    assert not jac_to_affine(S, bls.msm_g1(de, np.zeros((0, 4), dtype=np.uint64))).any()
    import davinci_node_amd as pkgmod
    with pytest.raises(pkgmod.DvnError):
        bls.msm_g1(de, np.zeros((6, 4), dtype=np.uint64))
    with pytest.raises(pkgmod.DvnError):
        ctx.msm_g1(de, np.zeros((5, 4), dtype=np.uint64))                       # BN254 entry point on a BLS12-377 set
    de.free(); dp.free()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [12, 14, 18])
def test_bls12377_msm_g1_closed_form(bls, ctx, S, log_n):
    """Device-generated P_i = P0 + i D: sum s_i P_i = (sum s_i) P0 + (sum i s_i) D with Python big ints."""
    n = 1 << log_n
    dp = bls.synthetic_g1(0xDA71C1, n)
    head = bls.download(dp, 0, 3); ref = S.synthetic_g1_points(0xDA71C1, 3)
    assert [list(r) for r in head] == [S.g1_to_limbs(q) for q in ref]
    p0, d = ref[0], S.g1_add(ref[1], S.g1_neg(ref[0]))
    tail = bls.download(dp, n - 1, 1)
    assert list(tail[0]) == S.g1_to_limbs(S.g1_add(p0, S.g1_mul(d, n - 1)))
    for kind in (0, 1):
        ds = bls.synthetic_scalars(1, kind, n)
        sm = np.empty((n, 4), dtype=np.uint64); ctx.d2h(sm, ds)
        so = sm.astype(object)
        vals = so[:, 0] + (so[:, 1] << 64) + (so[:, 2] << 128) + (so[:, 3] << 192)
        rinv = pow(1 << 256, -1, S.r)
        s0 = int(vals.sum()) * rinv % S.r
        s1 = int((vals * np.arange(n, dtype=object)).sum()) * rinv % S.r
        if kind == 0:
            assert S.fr_from_limbs(sm[5]) == S.fr_at(1, 5)                      # the stream is the oracle's
        want = S.g1_add(S.g1_mul(p0, s0), S.g1_mul(d, s1))
        got = jac_to_affine(S, bls.msm_g1(dp, ds, n=n))
        assert list(got) == S.g1_to_limbs(want), kind
        ctx.free(ds)
    dp.free()


@pytest.mark.gpu
def test_bls12377_msm_g2_vs_oracle(bls, S):
    q = S.synthetic_g2_points(4, 60)
    qa = np.array([S.g2_to_limbs(x) for x in q], dtype=np.uint64)
    dq = bls.upload_g2(qa)
    assert (bls.download(dq) == qa).all()
    for n in (1, 17, 60):
        ks = [S.fr_at(9, i) for i in range(n)]
        if n == 60:
            ks[0], ks[1], ks[2] = 0, S.r - 1, 1
        sc = np.array([S.fr_to_limbs(k) for k in ks], dtype=np.uint64)
        got = jac_to_affine(S, bls.msm_g2(dq, sc), g2=True)
        assert list(got) == S.g2_to_limbs(S.g2_msm(q[:n], ks)), n
    dq.free()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 9, 10])
def test_bls12377_ntt_all_modes_vs_oracle(bls, S, opy, log_n):
    n = 1 << log_n
    dom = bls.domain(log_n)
    d = S.domain(n)
    x = [S.fr_at(2, i) for i in range(n)]
    xm = np.array([S.fr_to_limbs(v) for v in x], dtype=np.uint64)
    for inv in (False, True):
        for dec in (0, 1):
            for cos in (False, True):
                y = list(x); (d.fft_inverse if inv else d.fft)(y, dec, cos)
                ym = xm.copy(); bls.ntt(dom, ym, inv, dec, cos)
                assert [S.fr_from_limbs(r) for r in ym] == y, (log_n, inv, dec, cos)
    dom.free()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,ln", [(3, 8), (6, 50), (9, 512)])
def test_bls12377_compute_h_vs_oracle(bls, S, log_n, ln):
    """computeH (3 iFFT, 3 coset FFT, pointwise, coset iFFT) on the curve's scalar field, ragged lengths zero-padded."""
    dom = bls.domain(log_n); d = S.domain(1 << log_n)
    a, b, c = ([S.fr_at(s, i) for i in range(ln)] for s in (11, 12, 13))
    L = lambda v: np.array([S.fr_to_limbs(x) for x in v], dtype=np.uint64)
    got = bls.compute_h(dom, L(a), L(b), L(c))
    assert [S.fr_from_limbs(r) for r in got] == d.compute_h(a, b, c)
    dom.free()


@pytest.mark.gpu
def test_bls12377_ntt_large_properties(bls, ctx, S, pkg):
    log_n = 20
    n = 1 << log_n
    dom = bls.domain(log_n)
    dx = bls.synthetic_scalars(2, 0, n)
    x = np.empty((n, 4), dtype=np.uint64); ctx.d2h(x, dx)
    y = np.empty_like(x)
    for cos in (False, True):
        bls.ntt(dom, dx, False, pkg.DIF, cos); bls.ntt(dom, dx, True, pkg.DIT, cos)
        ctx.d2h(y, dx); assert (y == x).all()
        bls.ntt(dom, dx, True, pkg.DIF, cos); bls.ntt(dom, dx, False, pkg.DIT, cos)
        ctx.d2h(y, dx); assert (y == x).all()
    one = np.array(S.fr_to_limbs(1), dtype=np.uint64)
    delta = np.zeros_like(x); delta[1] = one
    ctx.h2d(dx, delta); bls.ntt(dom, dx, False, pkg.DIF, False); ctx.d2h(y, dx)
    w = pow(S.root, 1 << (S.two_adicity - log_n), S.r)
    assert S.fr_from_limbs(y[0]) == 1 and S.fr_from_limbs(y[1]) == S.r - 1          # w^0, w^(n/2)
    assert S.fr_from_limbs(y[2]) == pow(w, n // 4, S.r) and S.fr_from_limbs(y[3]) == pow(w, 3 * n // 4, S.r)
    ctx.free(dx); dom.free()
    with pytest.raises(pkg.DvnError):
        ctx.ntt(dom if False else bls.domain(4), np.zeros((16, 4), dtype=np.uint64), False, 0)   # BN254 entry on a 377 domain
