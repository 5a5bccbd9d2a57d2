"""BW6-761 (the aggregator curve): the MSM kernels instantiated on 27-limb base-field elements and
377-bit scalars, checked against the big-integer oracle (oracle/curves_py.py Bw6761).  Parity unpinned by the
reference (no vectors); the constants are cross-checked in test_bw6761_constants."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "davinci-node_amd", "csrc")


@pytest.fixture(scope="module")
def W():
    from oracle import curves_py
    return curves_py.BW6_761


def affine(j):
    """Library output is the affine normal form with Z = 1 (or Z = 0 for infinity): strip Z."""
    return j[:24].copy() if j[24:].any() else np.zeros(24, dtype=np.uint64)


def _is_probable_prime(n):
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2; s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_bw6761_constants(W):
    from oracle import curves_py
    assert W.p.bit_length() == 761 and W.r == curves_py.BLS12_377.p and W.r.bit_length() == 377
    assert _is_probable_prime(W.p) and W.p % 4 == 3
    assert pow(W.p, 6, W.r) == 1 and all(pow(W.p, k, W.r) != 1 for k in (1, 2, 3))       # embedding degree 6
    # points used by the tests are arbitrary curve points (not cofactor-cleared): the MSM works on integer scalars
    for group in (1, 2):
        q = W.some_point(group)
        assert W.on_curve(group, q) and W.on_curve(group, W.add(q, q)) and W.on_curve(group, W.mul(q, 0xABCDEF))
    # scalar stream: six words, reduced mod r
    assert W.fr_at(1, 0) < W.r and W.fr_from_limbs(W.fr_to_limbs(W.fr_at(1, 5))) == W.fr_at(1, 5)


def test_bw6761_host_arithmetic(W, opy):
    """The 27-limb field and the group law (outlined multiplication, non-lazy accumulator) compiled for the host."""
    subprocess.check_call(["make", "-C", CSRC, "-s", "libdvn_hostcheck.so"])
    lib = C.CDLL(os.path.join(CSRC, "libdvn_hostcheck.so"))
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    rng = opy.SplitMix64(11)
    def rnd(mod):
        v = 0
        for i in range(12):
            v |= rng.next() << (64 * i)
        return v % mod
    L = lambda x: np.array(W.limbs(W.fp_to_mont(x), 12), dtype=np.uint64)
    for it in range(400):
        a, b = rnd(W.p), rnd(W.p)
        if it == 0: a = 0
        if it == 1: a = b = W.p - 1
        if it == 2: a, b = 1, W.p - 1
        am, bm, r = L(a), L(b), np.zeros(12, dtype=np.uint64)
        for op, w in {0: a * b % W.p, 1: (a + b) % W.p, 2: (a - b) % W.p, 3: a * a % W.p, 5: (-a) % W.p, 6: 2 * a % W.p}.items():
            lib.hc761_fp_op(p(am), p(bm), p(r), op)
            assert opy.from_limbs(r) == W.fp_to_mont(w), (op, hex(a), hex(b))
        if it < 4:
            lib.hc761_fp_op(p(am), p(bm), p(r), 4)
            assert opy.from_limbs(r) == W.fp_to_mont(pow(a, -1, W.p) if a else 0)
    for k in [0, 1, W.r - 1, rnd(W.r), rnd(W.r)]:                               # scalar image -> canonical words
        km = np.array(W.fr_to_limbs(k), dtype=np.uint64); r = np.zeros(6, dtype=np.uint64)
        lib.hc761_scalar_ext_to_canonical(p(km), p(r))
        assert opy.from_limbs(r) == k
    for group in (1, 2):
        pts = W.progression(W.some_point(group), 9, 6)
        ks = [W.fr_at(3, i) for i in range(6)]; ks[2] = 0; ks[3] = 1; ks[4] = W.r - 1
        pa = np.array([W.to_limbs(q) for q in pts], dtype=np.uint64)
        ka = np.array([W.limbs(k, 6) for k in ks], dtype=np.uint64)
        r = np.zeros(24, dtype=np.uint64); lib.hc761_msm_naive(p(pa), p(ka), C.c_size_t(6), p(r))
        assert list(r) == W.to_limbs(W.msm(pts, ks))
        for idx in ([0, 1, 2, 3], [5, 5], [5, 5 | (1 << 31)], [5, 5 | (1 << 31), 4], [3] * 9):
            ia = np.array(idx, dtype=np.uint32); r = np.zeros(24, dtype=np.uint64)
            lib.hc761_acc_chain(p(pa), p(ia), C.c_size_t(len(idx)), p(r))
            acc = None
            for e in idx:
                pt = pts[e & 0x7fffffff]; pt = W.neg(pt) if e >> 31 else pt; acc = W.add(acc, pt)
            assert list(r) == W.to_limbs(acc), idx


@pytest.fixture(scope="module")
def bw6(ctx, pkg):
    return pkg.Bw6761(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("group", [1, 2])
def test_bw6761_msm_vs_oracle(bw6, ctx, W, pkg, group):
    pts = W.progression(W.some_point(group), 0xDA71C1, 120)
    pa = np.array([W.to_limbs(q) for q in pts], dtype=np.uint64)
    pa[7] = 0; pts[7] = None                                                    # an infinity point
    dp = bw6.upload(group, pa)
    assert (bw6.download(dp) == pa).all()
    for n in (1, 2, 33, 120):
        ks = [W.fr_at(1 + n, i) for i in range(n)]
        if n >= 33:
            ks[3], ks[4], ks[5] = 0, 1, W.r - 1
        sc = np.array([W.fr_to_limbs(k) for k in ks], dtype=np.uint64)
        got = affine(bw6.msm(dp, sc))
        assert list(got) == W.to_limbs(W.msm(pts[:n], ks)), n
    # P and -P in one bucket, a repeated point, empty input, length mismatch, wrong curve
    edge = [pts[1], W.neg(pts[1]), pts[2], pts[2], pts[2]]
    de = bw6.upload(group, np.array([W.to_limbs(q) for q in edge], dtype=np.uint64))
    ks = [9, 9, 5, 5, 5]
    got = affine(bw6.msm(de, np.array([W.fr_to_limbs(k) for k in ks], dtype=np.uint64)))
    assert list(got) == W.to_limbs(W.msm(edge, ks))
    assert not affine(bw6.msm(de, np.zeros((0, 6), dtype=np.uint64))).any()
    with pytest.raises(pkg.DvnError):
        bw6.msm(de, np.zeros((6, 6), dtype=np.uint64))
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1(de, np.zeros((5, 4), dtype=np.uint64))                       # BN254 entry point on a BW6-761 set
    with pytest.raises(pkg.DvnError):
        bw6.upload(3, pa)
    de.free(); dp.free()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [13, 17])
def test_bw6761_msm_closed_form(bw6, ctx, W, log_n):
    """Device-generated P_i = P0 + i D: sum s_i P_i = (sum s_i) P0 + (sum i s_i) D with Python big ints (integer
    scalars: nothing is reduced mod r, the base point is not in the r-torsion)."""
    n = 1 << log_n
    base = W.some_point(1)
    dp = bw6.synthetic(1, np.array(W.to_limbs(base), dtype=np.uint64), 0xDA71C1, n)
    head = bw6.download(dp, 0, 3); ref = W.progression(base, 0xDA71C1, 3)
    assert [list(r) for r in head] == [W.to_limbs(q) for q in ref]
    p0, d = ref[0], W.add(ref[1], W.neg(ref[0]))
    tail = bw6.download(dp, n - 1, 1)
    assert list(tail[0]) == W.to_limbs(W.add(p0, W.mul(d, n - 1)))
    rinv = pow(1 << 384, -1, W.r)
    for kind in (0, 1):
        ds = bw6.synthetic_scalars(1, kind, n)
        sm = np.empty((n, 6), dtype=np.uint64); ctx.d2h(sm, ds)
        so = sm.astype(object)
        mont = so[:, 0] + (so[:, 1] << 64) + (so[:, 2] << 128) + (so[:, 3] << 192) + (so[:, 4] << 256) + (so[:, 5] << 320)
        vals = np.array([int(v) * rinv % W.r for v in mont], dtype=object)      # canonical INTEGER scalars
        want_stream = W.fr_at if kind == 0 else W.witness_like_scalar
        assert [int(v) for v in vals[:40]] == [want_stream(1, i) for i in range(40)]   # the stream is the oracle's
        s0 = int(vals.sum()); s1 = int((vals * np.arange(n, dtype=object)).sum())
        want = W.add(W.mul(p0, s0), W.mul(d, s1))
        got = affine(bw6.msm(dp, ds, n=n))
        assert list(got) == W.to_limbs(want), kind
        ctx.free(ds)
    dp.free()


def test_bw6761_scalar_domain_constants(W, opy):
    assert (W.r - 1) % (1 << 46) == 0 and (W.r - 1) % (1 << 47) != 0
    assert pow(15, (W.r - 1) // 2, W.r) == W.r - 1                               # 15 is a non-residue: it can generate
    assert W.root == 32863578547254505029601261939868325669770508939375122462904745766352256812585773382134936404344547323199885654433   # gnark-crypto fr/fft
    assert pow(W.root, 1 << 45, W.r) == W.r - 1
    d = W.domain(8); a = [W.fr_at(7, i) for i in range(8)]
    ref = [sum(a[j] * pow(d.gen, i * j, W.r) for j in range(8)) % W.r for i in range(8)]
    x = list(a); d.fft(x, opy.DIF)
    assert [x[opy.bitrev(i, 3)] for i in range(8)] == ref


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 9, 10])
def test_bw6761_ntt_all_modes_vs_oracle(bw6, W, opy, log_n):
    n = 1 << log_n
    dom = bw6.domain(log_n)
    d = W.domain(n)
    x = [W.fr_at(2, i) for i in range(n)]
    xm = np.array([W.fr_to_limbs(v) for v in x], dtype=np.uint64)
    for inv in (False, True):
        for dec in (0, 1):
            for cos in (False, True):
                y = list(x); (d.fft_inverse if inv else d.fft)(y, dec, cos)
                ym = xm.copy(); bw6.ntt(dom, ym, inv, dec, cos)
                assert [W.fr_from_limbs(r) for r in ym] == y, (log_n, inv, dec, cos)
    dom.free()


@pytest.mark.gpu
def test_bw6761_ntt_large_roundtrip_and_compute_h(bw6, ctx, W, pkg):
    log_n = 18
    n = 1 << log_n
    dom = bw6.domain(log_n)
    dx = bw6.synthetic_scalars(2, 0, n)
    x = np.empty((n, 6), dtype=np.uint64); ctx.d2h(x, dx)
    y = np.empty_like(x)
    for cos in (False, True):
        bw6.ntt(dom, dx, False, pkg.DIF, cos); bw6.ntt(dom, dx, True, pkg.DIT, cos)
        ctx.d2h(y, dx); assert (y == x).all()
        bw6.ntt(dom, dx, True, pkg.DIF, cos); bw6.ntt(dom, dx, False, pkg.DIT, cos)
        ctx.d2h(y, dx); assert (y == x).all()
    ctx.free(dx); dom.free()
    # computeH (ragged inputs) against the suite-generic oracle
    for lg, ln in ((3, 8), (6, 50), (9, 512)):
        dom = bw6.domain(lg); d = W.domain(1 << lg)
        a, b, c = ([W.fr_at(s, i) for i in range(ln)] for s in (11, 12, 13))
        L = lambda v: np.array([W.fr_to_limbs(t) for t in v], dtype=np.uint64)
        got = bw6.compute_h(dom, L(a), L(b), L(c))
        assert [W.fr_from_limbs(r) for r in got] == d.compute_h(a, b, c)
        dom.free()
    with pytest.raises(pkg.DvnError):
        ctx.ntt(bw6.domain(4), np.zeros((16, 4), dtype=np.uint64), False, 0)     # BN254 entry on a BW6-761 domain
