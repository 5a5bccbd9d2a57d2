"""The device arithmetic headers (9x29-bit Montgomery field, XYZZ group law), compiled for the host with g++
(csrc/libdvn_hostcheck.so) and checked against the oracle: the same code the HIP kernels run, minus the GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "davinci-node_amd", "csrc")


@pytest.fixture(scope="module")
def hc():
    subprocess.check_call(["make", "-C", CSRC, "-s", "libdvn_hostcheck.so"])
    return C.CDLL(os.path.join(CSRC, "libdvn_hostcheck.so"))


def p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


@pytest.mark.parametrize("field", ["fp", "fr"])
def test_field_ops(hc, opy, field):
    o = opy
    mod = o.P if field == "fp" else o.R
    fn = hc.hc_fp_op if field == "fp" else hc.hc_fr_op
    L = lambda x: np.array(o.limbs(x), dtype=np.uint64)
    V = lambda a: o.from_limbs(a)
    rng = o.SplitMix64(5)
    specials = [(0, 0), (0, 1), (mod - 1, mod - 1), (1, mod - 1), (mod - 1, 1), (2, mod - 2), ((1 << 253) % mod, (1 << 252) % mod)]
    for it in range(2000):
        a, b = specials[it] if it < len(specials) else (rng.next_fr() * 3 % mod, rng.next_fr() * 5 % mod)
        am, bm, r = L(o.to_mont(a, mod)), L(o.to_mont(b, mod)), np.zeros(4, dtype=np.uint64)
        exp = {0: a * b % mod, 1: (a + b) % mod, 2: (a - b) % mod, 3: a * a % mod, 5: (-a) % mod, 6: 2 * a % mod}
        for op, want in exp.items():
            fn(p(am), p(bm), p(r), op)
            assert V(r) == o.to_mont(want, mod), (field, op, hex(a), hex(b))
        if it < 12:
            fn(p(am), p(bm), p(r), 4)
            assert V(r) == o.to_mont(pow(a, -1, mod) if a else 0, mod)
        if field == "fr":
            hc.hc_fr_ext_to_canonical(p(am), p(r)); assert V(r) == a
        hc.hc_pack_roundtrip(p(am), p(r)); assert V(r) == V(am)
        assert hc.hc_fp_is_zero_of_diff(p(am), p(am)) == 1


def test_group_law_edge_cases(hc, opy, orc):
    o = opy
    pts = orc.g1_synthetic_points(9, 8)
    inf = np.zeros(8, dtype=np.uint64)
    neg = lambda pt: np.array(o.g1_to_limbs(o.g1_neg(o.g1_from_limbs([int(v) for v in pt]))), dtype=np.uint64)
    cases = [(pts[0], pts[1]), (pts[0], pts[0]), (pts[0], neg(pts[0])), (inf, pts[2]), (pts[2], inf), (inf, inf)]
    for a, b in cases:
        r = np.zeros(8, dtype=np.uint64)
        want = orc.g1_add(a, b)
        hc.hc_g1_add(p(a), p(b), p(r)); assert (r == want).all()
        hc.hc_g1_add_full(p(a), p(b), p(r)); assert (r == want).all()
    pts2 = orc.g2_synthetic_points(4, 3)
    r2 = np.zeros(16, dtype=np.uint64)
    for a, b in ((pts2[0], pts2[1]), (pts2[0], pts2[0])):
        hc.hc_g2_add(p(a), p(b), p(r2)); assert (r2 == orc.g2_add(a, b)).all()


def test_naive_msm_through_device_group_law(hc, opy, orc):
    o = opy
    pts = orc.g1_synthetic_points(9, 8)
    sc = [o.fr_at(3, i) for i in range(8)]; sc[3] = 0; sc[4] = 1; sc[5] = o.R - 1
    k = np.array([o.limbs(s) for s in sc], dtype=np.uint64)
    km = np.array([o.limbs(o.to_mont(s, o.R)) for s in sc], dtype=np.uint64)
    r = np.zeros(8, dtype=np.uint64)
    hc.hc_g1_msm_naive(p(pts), p(k), C.c_size_t(8), p(r))
    assert (r == orc.msm_g1(pts, km)).all()
    pts2 = orc.g2_synthetic_points(4, 5)
    r2 = np.zeros(16, dtype=np.uint64)
    hc.hc_g2_msm_naive(p(pts2), p(k), C.c_size_t(5), p(r2))
    assert (r2 == orc.msm_g2(pts2, km[:5])).all()


@pytest.mark.parametrize("which,name", [(0, "bls12377"), (1, "bls12381"), (2, "bw6761")])
def test_hostfast_wide_fields(hc, which, name):
    """The 64-bit-limb host finishing arithmetic of the wider curves (host_fast.h OpsFqN / OpsFq2N: window combine, blinding and
    assembly of a proof, KZG result compression) against Python big ints."""
    import random
    from oracle import curves_py as cp
    p_ = {0: cp.BLS12_377.p, 1: cp.BLS12_381.p, 2: cp.BW6_761.p}[which]
    nw = 12 if which == 2 else 6
    nl = 27 if which == 2 else 14
    Rm = 1 << (64 * nw)
    enc = lambda x: np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nw)], dtype=np.uint64)
    dec = lambda a: sum(int(v) << (64 * i) for i, v in enumerate(a))
    q = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
    rng = random.Random(which + 1)
    vals = [(0, 0), (1, p_ - 1), (p_ - 1, p_ - 1)] + [(rng.randrange(p_), rng.randrange(p_)) for _ in range(200)]
    r = np.zeros(nw, dtype=np.uint64)
    for i, (a, b) in enumerate(vals):
        am, bm = enc(a * Rm % p_), enc(b * Rm % p_)
        for op, want in ((0, a * b % p_), (1, (a + b) % p_), (2, (a - b) % p_)):
            hc.hc_hostfast_op(which, q(am), q(bm), q(r), op)
            assert dec(r) == want * Rm % p_, (name, op, i)
        if i < 8 and a:
            hc.hc_hostfast_op(which, q(am), q(bm), q(r), 4)
            assert dec(r) == pow(a, -1, p_) * Rm % p_
        # device image x * 2^(29 NL) -> x * 2^(64 NW)
        img = np.frombuffer((a * (1 << (29 * nl)) % p_).to_bytes(8 * nw, "little"), dtype=np.uint32).copy()
        hc.hc_hostfast_from_device(which, img.ctypes.data_as(C.POINTER(C.c_uint32)), q(r))
        assert dec(r) == a * Rm % p_
    if which < 2:
        beta = 5 if which == 0 else 1
        r2 = np.zeros(12, dtype=np.uint64)
        for _ in range(50):
            a0, a1, b0, b1 = (rng.randrange(p_) for _ in range(4))
            A = np.concatenate([enc(a0 * Rm % p_), enc(a1 * Rm % p_)]); B = np.concatenate([enc(b0 * Rm % p_), enc(b1 * Rm % p_)])
            hc.hc_hostfast_fp2_mul(which, q(A), q(B), q(r2), 0)
            assert (dec(r2[:6]), dec(r2[6:])) == ((a0 * b0 - beta * a1 * b1) % p_ * Rm % p_, (a0 * b1 + a1 * b0) % p_ * Rm % p_)
            hc.hc_hostfast_fp2_mul(which, q(A), q(B), q(r2), 1)
            d = pow((a0 * a0 + beta * a1 * a1) % p_, -1, p_)
            assert (dec(r2[:6]), dec(r2[6:])) == (a0 * d % p_ * Rm % p_, (-a1 * d) % p_ * Rm % p_)


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_carry_free_butterfly_forms(hc, field):
    """The forms the NTT butterflies use instead of carry passes and conditional subtractions, on raw limb arrays:
    fe_sub_raw (a + K p - b with no negative limb), fe_normalise, a multiplication whose first operand is such a raw value,
    and fe_reduce_full (any value below 2^(BITS + 6) -> below 2p, exactly), on every scalar field the NTT is built for.
    Adversarial values: multiples of p and their neighbours, all-ones limbs, the largest value each form is specified for."""
    import random
    from oracle import curves_py as cp
    from oracle import bn254_py as o
    mod = {0: o.R, 1: cp.BLS12_377.r, 2: cp.BLS12_381.r, 3: cp.BLS12_377.p}[field]
    nl = 14 if field == 3 else 9
    bits = mod.bit_length()
    B = 1 << 29
    norm = lambda x: [(x >> (29 * i)) & (B - 1) for i in range(nl - 1)] + [x >> (29 * (nl - 1))]
    val = lambda l: sum(int(v) << (29 * i) for i, v in enumerate(l))
    arr = lambda l: np.array(l, dtype=np.uint32)
    out = np.zeros(nl, dtype=np.uint32)
    zero = arr([0] * nl)

    def call(op, a, b=None):
        assert hc.hc_raw_op(field, op, p(arr(a)), p(arr(b) if b is not None else zero), p(out)) == nl
        return [int(v) for v in out]

    rng = random.Random(field)
    # fe_reduce_full: x < 2^(bits + 6)
    top = 1 << (bits + 6)
    xs = [0, 1, mod - 1, mod, mod + 1, 2 * mod - 1, 2 * mod, top - 1, top - mod, (top // mod) * mod, (top // mod) * mod - 1]
    xs += [k * mod + d for k in (1, 2, 3, 7, 18, 19, 31, 63) for d in (-1, 0, 1) if 0 <= k * mod + d < top]
    xs += [rng.randrange(top) for _ in range(3000)] + [rng.randrange(20) * mod + rng.randrange(1 << 40) for _ in range(500)]
    for x in xs:
        got = call(0, norm(x))
        assert all(v < B for v in got) and val(got) < 2 * mod and val(got) % mod == x % mod and (x - val(got)) % mod == 0, hex(x)
        assert x - val(got) in ((x // mod) * mod, (x // mod - 1) * mod)
        assert val(call(6, norm(x))) == x % mod
    # fe_sub_raw<K, SH>: b's limbs < 2^SH, b's value at most K p / 2 (the callers' bound), a anything with limbs < 2^30
    for op, K, SH in ((1, 8, 29), (2, 16, 30), (3, 2, 29)):
        bmax = K * mod // 2
        for it in range(1500):
            a = [rng.randrange(1 << 30) for _ in range(nl - 1)] + [rng.randrange(1 << 20)]
            if it % 3 == 0:
                bl = norm(rng.randrange(bmax))
            elif it % 3 == 1:
                bl = norm(bmax - 1 - rng.randrange(1 << 30))
            else:                                              # un-normalised subtrahend: a sum of two normalised values
                b1, b2 = rng.randrange(bmax // 2), rng.randrange(bmax // 2)
                bl = [u + v for u, v in zip(norm(b1), norm(b2))] if SH == 30 else norm(b1 + b2)
            if it == 5 and norm(bmax)[-1] >= 1:
                bl = [(1 << SH) - 1] * (nl - 1) + [norm(bmax)[-1] - 1]     # every limb at its bound
            got = call(op, a, bl)
            assert val(got) == val(a) + K * mod - val(bl), (op, it)          # exact as integers: no limb wrapped
            assert all(v < (1 << 30) + (1 << SH) + B for v in got)
            n = call(4, got)
            assert val(n) == val(got) and all(v < B for v in n[:-1])
    # a raw first operand (limbs up to 2.5 * 2^30 on 9-limb fields, 1.5 * 2^30 on wider ones) times a canonical one
    lim = (5 << 29) if nl == 9 else (3 << 29)
    Rinv = pow(1 << (29 * nl), -1, mod)
    for it in range(1500):
        a = [lim - 1 - (rng.randrange(4) if it % 2 else rng.randrange(lim)) for _ in range(nl - 1)] + [rng.randrange(1 << 26)]
        if val(a) >= 24 * mod:
            a[-1] = rng.randrange(max(1, (20 * mod) >> (29 * (nl - 1))))
        w = norm(mod - 1 - rng.randrange(1 << 50) if it % 2 else rng.randrange(mod))
        got = call(5, a, w)
        assert all(v < B for v in got[:-1]) and val(got) < 2 * mod and val(got) % mod == val(a) * val(w) * Rinv % mod, it


def test_lazy_accumulator_chain_with_signs(hc, opy, orc):
    """k_accumulate's loop body on the host: the lazily reduced XYZZ accumulator with the entry's sign folded into the
    addition (acc_add_mixed_signed), against the oracle's affine group law -- plain sums, P + P (doubling branch), P - P
    (infinity, then restart), a negated first entry, -P - P, and long random chains that keep the accumulator loose."""
    import random
    o = opy
    pts = orc.g1_synthetic_points(21, 16)
    aff = [o.g1_from_limbs([int(v) for v in r]) for r in pts]
    rng = random.Random(7)
    N = 1 << 31
    cases = [[0, 1, 2, 3], [5, 5], [5, 5 | N], [5, 5 | N, 7], [5 | N], [5 | N, 5 | N], [5 | N, 5], [3] * 20, [3 | N] * 9 + [3] * 9 + [4 | N]]
    cases += [[rng.randrange(16) | (N if rng.random() < 0.5 else 0) for _ in range(rng.randrange(2, 200))] for _ in range(40)]
    for idx in cases:
        ia = np.array(idx, dtype=np.uint32); r = np.zeros(8, dtype=np.uint64)
        hc.hc_g1_acc_chain(p(pts), p(ia), C.c_size_t(len(idx)), p(r))
        acc = None
        for e in idx:
            pt = aff[e & 0x7fffffff]
            acc = o.g1_add(acc, o.g1_neg(pt) if e >> 31 else pt)
        assert [int(v) for v in r] == o.g1_to_limbs(acc), idx[:8]


@pytest.mark.parametrize("kh,kl", [(1, 1), (3, 2), (5, 5), (10, 3), (2, 10)])
def test_one_bucket_set_reduction_host_half(hc, opy, orc, kh, kl):
    """sum_b (b + 1) B_b through row / column sums, subset sums by weight bit and the host's combine (host_fast.h
    bitsum_combine_pieces, the code msm_collect runs on the device's pieces) equals the plain weighted sum: buckets B_b = [k_b] G with
    zeros, ones and random multipliers; 2^10 rows or columns give two pieces per bit."""
    o = opy
    nb = 1 << (kh + kl)
    rng = np.random.default_rng(kh * 100 + kl)
    ks = rng.integers(0, 1 << 20, size=nb, dtype=np.uint32)
    ks[::7] = 0; ks[1::11] = 1
    out = np.zeros(12, dtype=np.uint64)
    hc.hc_bitsum_reduction(C.c_uint32(kh), C.c_uint32(kl), p(ks), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    total = sum((b + 1) * int(k) for b, k in enumerate(ks)) % o.R
    g = np.array([o.limbs(o.to_mont(1, o.P)) + o.limbs(o.to_mont(2, o.P))], dtype=np.uint64)
    want = orc.msm_g1(g, np.array([o.limbs(o.to_mont(total, o.R))], dtype=np.uint64))
    assert (out[:8] == np.asarray(want).reshape(-1)[:8]).all()


def test_ntt_pass_plan(hc):
    """Stage counts of the passes of a transform: they add up to log_n, none exceeds the tile limit, and the pass that holds the last
    two stages (last of DIF, first of DIT) has an even count whenever the transform has more than one pass."""
    ks = (C.c_uint32 * 16)()
    for kmax in (8, 10, 11):
        for log_n in range(1, 28):
            for dif in (1, 0):
                npass = hc.hc_ntt_pass_plan(log_n, kmax, dif, ks)
                got = list(ks[:npass])
                assert npass == -(-log_n // kmax) and sum(got) == log_n and max(got) <= kmax and min(got) >= 1, (log_n, kmax, dif, got)
                if npass > 1 and kmax % 2 == 0:          # (an odd limit can leave every pass full and odd: 11 + 11)
                    end = got[-1] if dif else got[0]
                    assert end % 2 == 0, (log_n, kmax, dif, got)
    hc.hc_ntt_pass_plan(24, 10, 1, ks); assert list(ks[:3]) == [8, 8, 8]
    hc.hc_ntt_pass_plan(20, 10, 1, ks); assert list(ks[:2]) == [10, 10]
    hc.hc_ntt_pass_plan(21, 10, 1, ks); assert list(ks[:3]) == [8, 7, 6]
    hc.hc_ntt_pass_plan(21, 10, 0, ks); assert list(ks[:3]) == [6, 8, 7]


def test_expand_plan(hc):
    """Window size of a set's expanded table: 15 bits from 2^10 points, 16 from 2^15, 17 from 2^18, 20 from 2^20 -- moved up to the next
    size whose top row holds at least 8 bits of the scalar (254-, 253-, 255- and 377-bit fields) -- and no table above the size limit."""
    c, rows = C.c_uint32(), C.c_uint32()
    def plan(n, bits, pb=64, min_log=10, max_bytes=16 << 30):
        hc.hc_expand_plan(C.c_uint64(n), bits, C.c_uint64(pb), min_log, C.c_uint64(max_bytes), C.byref(c), C.byref(rows))
        return c.value, rows.value
    assert plan(1023, 254) == (0, 1)
    assert plan(1 << 10, 254) == (15, 17) and plan((1 << 15) - 1, 254) == (15, 17)
    assert plan(1 << 15, 254) == (16, 16) and plan(1 << 18, 254) == (17, 15) and plan(1 << 20, 254) == (20, 13) and plan(1 << 24, 254) == (20, 13)
    assert plan(1 << 14, 255) == (16, 16)            # 255 = 15 * 17: a 15-bit window would leave the top row nothing but carries
    assert plan(1 << 18, 255) == (19, 14)            # 255 % 17 = 0, 255 % 18 = 3 -> 19
    assert plan(1 << 18, 377, 192) == (18, 21)       # 377 = 22 * 17 + 3
    for bits in (253, 254, 255, 377):
        for ln in range(10, 25):
            cc, rr = plan(1 << ln, bits)
            if cc:
                assert bits % cc >= 8 or cc == 21, (bits, ln, cc)
                assert rr == bits // cc + 1
    assert plan(1 << 24, 254, 128, 10, 16 << 30) == (0, 1)      # 13 rows x 2^24 x 128 B = 26 GiB: stays plain
    assert plan(1 << 27, 254) == (0, 1)                          # 13 x 2^27 entry positions do not fit 31 bits
