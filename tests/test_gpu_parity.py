"""Parity of the HIP path (through the C-ABI, include/dvn_prover.h) against the oracle.  Bit-exact everywhere:
this is integer work.  MSM results are compared after affine normalisation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def aff1(orc, jac):
    return orc.g1_jac_to_affine(jac)


def aff2(orc, jac):
    return orc.g2_jac_to_affine(jac)


# ---- point sets -------------------------------------------------------------------------------------------------
def test_points_roundtrip_and_synthetic(ctx, orc):
    ph = orc.g1_synthetic_points(3, 300)
    ph[7] = 0                                           # an infinity point survives the internal form
    pu = ctx.upload_g1(ph)
    assert len(pu) == 300 and (pu.download() == ph).all()
    assert (pu.download(10, 5) == ph[10:15]).all()
    pu.free()
    for n in (1, 5, 1000):
        ps = ctx.synthetic_g1(0xDA71C1, n)
        assert (ps.download() == orc.g1_synthetic_points(0xDA71C1, n)).all()
        ps.free()
    p2 = ctx.synthetic_g2(7, 50)
    assert (p2.download() == orc.g2_synthetic_points(7, 50)).all()
    p2.free()


def test_synthetic_scalars_match_oracle_streams(ctx, orc):
    for kind in (0, 1):
        n = 5000
        d = ctx.synthetic_scalars(1, kind, n)
        got = np.empty((n, 4), dtype=np.uint64); ctx.d2h(got, d); ctx.free(d)
        assert (got == orc.scalars(1, kind, n)).all()


# ---- MSM G1 -----------------------------------------------------------------------------------------------------
def test_msm_g1_golden_vectors(ctx, orc, vectors, conv):
    for case in vectors["msm_g1"]:
        pts = ctx.upload_g1(conv.g1(case["points"]))
        got = aff1(orc, ctx.msm_g1(pts, conv.fr(case["scalars"])))
        assert (got == conv.g1([case["result"]])[0]).all(), case["name"]
        pts.free()


# 2^20 is BASELINE config 2 ("2^20 random scalars/points, bit-exact vs MultiExp"), (1 << 21) + 5 a ragged expanded set
@pytest.mark.parametrize("n", [1, 2, 33, 1000, 1 << 12, (1 << 14) + 17, 1 << 16, 1 << 18, 1 << 20, (1 << 21) + 5])
@pytest.mark.parametrize("kind", [0, 1])
def test_msm_g1_vs_oracle(ctx, orc, n, kind):
    ph = orc.g1_synthetic_points(0xDA71C1, n)
    pts = ctx.upload_g1(ph)
    sc = orc.scalars(1 + n, kind, n)
    assert (aff1(orc, ctx.msm_g1(pts, sc)) == orc.msm_g1(ph, sc)).all()
    pts.free()


def test_msm_g1_offsets_subranges_and_device_scalars(ctx, orc):
    n = 5000
    ph = orc.g1_synthetic_points(11, n)
    pts = ctx.upload_g1(ph)
    sc = orc.scalars(5, 0, n)
    for off, m in ((0, n), (17, 1000), (4999, 1), (1234, 3000)):
        assert (aff1(orc, ctx.msm_g1(pts, sc[:m], off=off)) == orc.msm_g1(ph[off:off + m], sc[:m])).all()
    d = ctx.to_device(sc)
    assert (aff1(orc, ctx.msm_g1(pts, d, n=n)) == orc.msm_g1(ph, sc)).all()
    ctx.free(d); pts.free()


def test_msm_g1_empty_and_errors(ctx, orc, pkg):
    ph = orc.g1_synthetic_points(11, 10)
    pts = ctx.upload_g1(ph)
    empty = ctx.msm_g1(pts, np.zeros((0, 4), dtype=np.uint64))
    assert (aff1(orc, empty) == 0).all()                                   # infinity
    with pytest.raises(pkg.DvnError) as e:                                 # MultiExp: len(points) != len(scalars)
        ctx.msm_g1(pts, orc.scalars(1, 0, 11))
    assert e.value.code == -1
    p2 = ctx.upload_g2(orc.g2_synthetic_points(1, 4))
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1(p2, orc.scalars(1, 0, 4))                               # wrong curve handle
    p2.free(); pts.free()


def test_msm_g1_hot_bucket_skew(ctx, orc):
    """All scalars equal (every entry of every window in one bucket) and all ones: the heavy-bucket path."""
    n = 1 << 15
    ph = orc.g1_synthetic_points(21, n)
    pts = ctx.upload_g1(ph)
    one = orc.scalars(1, 0, 1)
    for sc in (np.repeat(one, n, axis=0), np.repeat(orc.scalars(1, 1, 2000)[-1:], n, axis=0)):
        assert (aff1(orc, ctx.msm_g1(pts, sc)) == orc.msm_g1(ph, sc)).all()
    pts.free()


def test_msm_g1_linearity_and_window_sharding(ctx, orc, opy):
    n = 3000
    ph = orc.g1_synthetic_points(31, n)
    pts = ctx.upload_g1(ph)
    a, b = orc.scalars(1, 0, n), orc.scalars(2, 0, n)
    ab = np.array([orc.fr_add(a[i], b[i]) for i in range(n)], dtype=np.uint64)
    ra, rb, rab = (aff1(orc, ctx.msm_g1(pts, s)) for s in (a, b, ab))
    assert (orc.g1_add(ra, rb) == rab).all()
    # window-sharded evaluation: any split of the windows recombines to the same point
    _, nwin, cb = ctx.msm_plan_points(pts, n)
    for cuts in ([0, nwin], [0, nwin // 2, nwin], [0, 1, 5, nwin]):
        parts = [ctx.msm_g1_windows(pts, a, lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
        got = aff1(orc, ctx.msm_combine_g1(np.concatenate(parts), cb))
        assert (got == ra).all()
    pts.free()
    # the same on an EXPANDED point set (one bucket set, partial results add up: combine_bits = 0)
    n = 1 << 16
    ph = orc.g1_synthetic_points(32, n)
    pts = ctx.upload_g1(ph)
    a = orc.scalars(3, 0, n)
    want = orc.msm_g1(ph, a)
    assert ctx.msm_plan_points(pts, n) == (16, 16, 0)               # default policy: 2^15 <= n < 2^18 -> rows of 16-bit windows
    assert (aff1(orc, ctx.msm_g1(pts, a)) == want).all()
    pts.set_expanded(0)
    assert ctx.msm_plan_points(pts, n)[2] != 0                      # plain set: per-window sums need the Horner combine
    pts.set_expanded(18)
    assert (pts.download(100, 50) == ph[100:150]).all()             # row 0 is untouched
    cw, nwin, cb = ctx.msm_plan_points(pts, n)
    assert (cw, nwin, cb) == (18, 15, 0)
    assert (aff1(orc, ctx.msm_g1(pts, a)) == want).all()
    for cuts in ([0, nwin], [0, nwin // 2, nwin], [0, 1, 5, nwin]):
        parts = [ctx.msm_g1_windows(pts, a, lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
        assert (aff1(orc, ctx.msm_combine_g1(np.concatenate(parts), cb)) == want).all()
    # a short sub-range of an expanded set falls back to the classic per-window path
    assert (aff1(orc, ctx.msm_g1(pts, a[:1000], off=77)) == orc.msm_g1(ph[77:1077], a[:1000])).all()
    assert (aff1(orc, ctx.msm_g1(pts, a[:40000], off=5)) == orc.msm_g1(ph[5:40005], a[:40000])).all()
    for kind_scalars in (np.repeat(orc.scalars(1, 0, 1), n, axis=0), orc.scalars(9, 1, n)):      # hot bucket / witness-like
        assert (aff1(orc, ctx.msm_g1(pts, kind_scalars)) == orc.msm_g1(ph, kind_scalars)).all()
    pts.set_expanded(0)
    assert (aff1(orc, ctx.msm_g1(pts, a)) == want).all()
    pts.free()


@pytest.mark.parametrize("log_n", [20, 22])
def test_msm_g1_full_size_closed_form(ctx, orc, opy, log_n):
    """BASELINE sizes: MSM over P_i = P0 + i D equals (sum s_i) P0 + (sum i s_i) D, computed independently."""
    o = opy
    n = 1 << log_n
    pts = ctx.synthetic_g1(0xDA71C1, n)
    two = pts.download(0, 2)
    assert (two == orc.g1_synthetic_points(0xDA71C1, 2)).all()
    tail = pts.download(n - 3, 3)
    p0 = o.g1_from_limbs([int(v) for v in two[0]]); p1 = o.g1_from_limbs([int(v) for v in two[1]])
    d = o.g1_add(p1, o.g1_neg(p0))
    assert list(tail[2]) == o.g1_to_limbs(o.g1_add(p0, o.g1_mul(d, n - 1)))        # spot-check the generator
    for kind in (0, 1):
        sc = orc.scalars(1, kind, n)
        can = orc.fr_vec_from_mont(sc).astype(object)
        vals = can[:, 0] + (can[:, 1] << 64) + (can[:, 2] << 128) + (can[:, 3] << 192)
        s0 = int(vals.sum()) % R
        s1 = int((vals * np.arange(n, dtype=object)).sum()) % R
        want = o.g1_add(o.g1_mul(p0, s0), o.g1_mul(d, s1))
        assert list(aff1(orc, ctx.msm_g1(pts, sc))) == o.g1_to_limbs(want)
    pts.free()


def test_msm_async_in_flight(ctx, orc, pkg):
    """submit/collect: MSMs in flight on the context's four lanes, results identical to the synchronous call; a fifth
    submit is refused until a ticket is collected."""
    n = 20000
    ph = orc.g1_synthetic_points(41, n); pts = ctx.upload_g1(ph)
    a, b = orc.scalars(1, 0, n), orc.scalars(2, 1, n)
    da, db = ctx.to_device(a), ctx.to_device(b)
    for _ in range(3):
        t1 = ctx.msm_g1_submit(pts, da, n=n)
        t2 = ctx.msm_g1_submit(pts, db, n=n)
        t3 = ctx.msm_g1_submit(pts, da, n=n)
        t4 = ctx.msm_g1_submit(pts, db, n=n)
        assert {t1, t2, t3, t4} == {1, 2, 3, 4}
        with pytest.raises(pkg.DvnError):
            ctx.msm_g1_submit(pts, da, n=n)
        r4 = ctx.msm_g1_collect(t4); r2 = ctx.msm_g1_collect(t2); r1 = ctx.msm_g1_collect(t1); r3 = ctx.msm_g1_collect(t3)
        assert (aff1(orc, r1) == orc.msm_g1(ph, a)).all() and (aff1(orc, r2) == orc.msm_g1(ph, b)).all()
        assert (r3 == r1).all() and (r4 == r2).all()
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1_collect(1)                       # nothing outstanding
    # host scalars may be reused as soon as submit returns
    h = a.copy(); t1 = ctx.msm_g1_submit(pts, h); h[:] = 0
    assert (aff1(orc, ctx.msm_g1_collect(t1)) == orc.msm_g1(ph, a)).all()
    p2h = orc.g2_synthetic_points(3, 500); p2 = ctx.upload_g2(p2h); s2 = orc.scalars(5, 0, 500)
    t2 = ctx.msm_g2_submit(p2, s2); t1 = ctx.msm_g1_submit(pts, da, n=n)
    assert (aff1(orc, ctx.msm_g1_collect(t1)) == orc.msm_g1(ph, a)).all()
    assert (aff2(orc, ctx.msm_g2_collect(t2)) == orc.msm_g2(p2h, s2)).all()
    ctx.free(da); ctx.free(db); pts.free(); p2.free()


def test_window_sharded_pipeline_single_process(ctx, orc, pkg):
    """sharding.ShardedMsmPipeline with world = 1 semantics replaced by two fake ranks sharing this GPU: both ranks'
    window slices are submitted asynchronously (two lanes), collected, concatenated and combined."""
    from davinci_node_amd import sharding
    n = 30000
    ph = orc.g1_synthetic_points(51, n); pts = ctx.upload_g1(ph)
    sc = orc.scalars(7, 0, n); d = ctx.to_device(sc)
    _, nwin, cb = ctx.msm_plan_points(pts, n)
    want = orc.msm_g1(ph, sc)
    for world in (2, 4):
        parts = []
        slices = [sharding.window_slice(nwin, world, r) for r in range(world)]
        for i in range(0, world, 2):                       # two tickets in flight at a time
            t0 = ctx.msm_g1_windows_submit(pts, d, *slices[i], n=n)
            t1 = ctx.msm_g1_windows_submit(pts, d, *slices[i + 1], n=n)
            parts += [ctx.msm_g1_windows_collect(t0), ctx.msm_g1_windows_collect(t1)]
        assert (aff1(orc, ctx.msm_combine_g1(np.concatenate(parts), cb)) == want).all()
    ctx.free(d); pts.free()


def test_thread_safety_concurrent_callers(ctx, orc):
    """gnark issues its MSMs from concurrent goroutines and cgo calls arrive on arbitrary OS threads:
    four threads hammer one context with MSMs and NTTs; every result must still be bit-exact."""
    import threading
    n = 6000
    ph = orc.g1_synthetic_points(61, n); pts = ctx.upload_g1(ph)
    scal = [orc.scalars(100 + i, i % 2, n) for i in range(4)]
    want = [orc.msm_g1(ph, s) for s in scal]
    dom = ctx.domain(10); x = orc.scalars(2, 0, 1 << 10); wx = orc.ntt(x, False, 0, False)
    errors = []
    def worker(i):
        try:
            for _ in range(6):
                if not (aff1(orc, ctx.msm_g1(pts, scal[i])) == want[i]).all():
                    errors.append(("msm", i))
                y = x.copy(); ctx.ntt(dom, y, False, 0, False)
                if not (y == wx).all():
                    errors.append(("ntt", i))
        except Exception as e:                                    # noqa: BLE001
            errors.append((repr(e), i))
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in ths: th.start()
    for th in ths: th.join()
    assert errors == []
    dom.free(); pts.free()


def test_context_lifecycle_and_second_context(pkg, orc):
    """init/shutdown repeatedly; two live contexts on one device are independent."""
    for _ in range(3):
        c = pkg.Context(0); c.close()
    a, b = pkg.Context(0), pkg.Context(0)
    ph = orc.g1_synthetic_points(71, 500); sc = orc.scalars(3, 0, 500)
    pa, pb = a.upload_g1(ph), b.upload_g1(ph)
    ra, rb = a.msm_g1(pa, sc), b.msm_g1(pb, sc)
    assert (ra == rb).all() and (orc.g1_jac_to_affine(ra) == orc.msm_g1(ph, sc)).all()
    pa.free(); pb.free(); a.close(); b.close()


# ---- MSM G2 -----------------------------------------------------------------------------------------------------
def test_msm_g2_golden_vectors(ctx, orc, vectors, conv):
    for case in vectors["msm_g2"]:
        pts = ctx.upload_g2(conv.g2(case["points"]))
        got = aff2(orc, ctx.msm_g2(pts, conv.fr(case["scalars"])))
        assert (got == conv.g2([case["result"]])[0]).all(), case["name"]
        pts.free()


@pytest.mark.parametrize("n", [1, 17, 1000, 1 << 13, (1 << 17) + 3, 1 << 19])          # the last two are expanded sets
def test_msm_g2_vs_oracle(ctx, orc, n):
    ph = orc.g2_synthetic_points(5, n)
    pts = ctx.upload_g2(ph)
    for kind in (0, 1):
        sc = orc.scalars(9, kind, n)
        assert (aff2(orc, ctx.msm_g2(pts, sc)) == orc.msm_g2(ph, sc)).all()
    pts.free()


def test_msm_g2_every_bucket_heavy(ctx, orc):
    """Regression for the out-of-range LDS addressing of call-based reductions: with 11-bit windows on 2^16
    points every one of the 1024 buckets spans ~96 segments, so each block of k_fix_heavy iterates twice.  Plain and expanded
    evaluation must agree (and with the oracle on a prefix)."""
    n = 1 << 16
    pts = ctx.synthetic_g2(5, n)
    d = ctx.synthetic_scalars(1, 0, n)
    pts.set_expanded(0)
    ref = ctx.msm_g2(pts, d, n=n)
    pts.set_expanded(11)
    for _ in range(3):
        assert (ctx.msm_g2(pts, d, n=n) == ref).all()
    pts.set_expanded(0)
    sm = np.empty((2000, 4), dtype=np.uint64); ctx.d2h(sm, d)
    ph = pts.download(0, 2000)
    assert (aff2(orc, ctx.msm_g2(pts, sm)) == orc.msm_g2(ph, sm)).all()
    ctx.free(d); pts.free()


# ---- NTT ----------------------------------------------------------------------------------------------------------
def test_ntt_golden_vectors(ctx, vectors, conv):
    doms = {}
    for case in vectors["ntt"]:
        dom = doms.setdefault(case["log_n"], ctx.domain(case["log_n"]))
        x = conv.fr(case["input"])
        ctx.ntt(dom, x, bool(case["inverse"]), case["decimation"], bool(case["coset"]))
        assert (x == conv.fr(case["output"])).all(), {k: case[k] for k in ("log_n", "inverse", "decimation", "coset")}
    for d in doms.values():
        d.free()


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_ntt_all_modes_vs_oracle(ctx, orc, log_n):
    n = 1 << log_n
    dom = ctx.domain(log_n)
    x = orc.scalars(2, 0, n)
    for inv in (False, True):
        for dec in (0, 1):
            for cos in (False, True):
                y = x.copy()
                ctx.ntt(dom, y, inv, dec, cos)
                assert (y == orc.ntt(x, inv, dec, cos)).all(), (log_n, inv, dec, cos)
    dom.free()


def test_ntt_full_size_properties(ctx, orc, pkg):
    """2^24 (BASELINE config 3): device-resident round trips in every pairing + linearity, no CPU transform."""
    log_n = 24
    n = 1 << log_n
    dom = ctx.domain(log_n)
    x = orc.scalars(2, 0, n)
    dx = ctx.to_device(x)
    y = np.empty_like(x)
    for cos in (False, True):
        ctx.ntt(dom, dx, False, pkg.DIF, cos); ctx.ntt(dom, dx, True, pkg.DIT, cos)
        ctx.d2h(y, dx); assert (y == x).all()
        ctx.ntt(dom, dx, True, pkg.DIF, cos); ctx.ntt(dom, dx, False, pkg.DIT, cos)
        ctx.d2h(y, dx); assert (y == x).all()
    # a delta at index 1 transforms (DIF, bit-reversed output) to the powers of the generator: y[rev(k)] = w^k
    delta = np.zeros_like(x); delta[1] = orc.fr_to_mont(np.array([1, 0, 0, 0], dtype=np.uint64))
    ctx.h2d(dx, delta); ctx.ntt(dom, dx, False, pkg.DIF, False); ctx.d2h(y, dx)
    w = orc.ntt(np.array([[0, 0, 0, 0], delta[1]], dtype=np.uint64), False, 0, False)     # placeholder shape check
    assert w.shape == (2, 4)
    one = delta[1]
    assert (y[0] == one).all()                                    # w^0
    half = y[1]                                                   # rev(1<<23)=1 -> w^(n/2) = -1
    assert (orc.fr_add(half, one) == 0).all()
    quarter = y[2]                                                # rev(2) = n/4 -> w^(n/4), a 4th root of unity
    assert (orc.fr_mul(quarter, quarter) == half).all()
    # and the transform itself, bit for bit against the CPU restatement: forward DIF and inverse DIT on the coset
    ctx.h2d(dx, x); ctx.ntt(dom, dx, False, pkg.DIF, False); ctx.d2h(y, dx)
    assert (y == orc.ntt(x, False, pkg.DIF, False)).all()
    ctx.ntt(dom, dx, True, pkg.DIT, True); ctx.d2h(y, dx)
    assert (y == orc.ntt(orc.ntt(x, False, pkg.DIF, False), True, pkg.DIT, True)).all()
    # the two coset kinds whose shift lives in the twiddle tables (what computeH runs): FFT(DIT, coset) and FFTInverse(DIF, coset)
    ctx.h2d(dx, x); ctx.ntt(dom, dx, False, pkg.DIT, True); ctx.d2h(y, dx)
    want = orc.ntt(x, False, pkg.DIT, True)
    assert (y == want).all()
    ctx.ntt(dom, dx, True, pkg.DIF, True); ctx.d2h(y, dx)
    assert (y == orc.ntt(want, True, pkg.DIF, True)).all()
    ctx.free(dx); dom.free()


# ---- computeH --------------------------------------------------------------------------------------------------
def test_compute_h_golden_vectors(ctx, vectors, conv):
    for case in vectors["compute_h"]:
        dom = ctx.domain(case["log_n"])
        h = ctx.compute_h(dom, conv.fr(case["a"]), conv.fr(case["b"]), conv.fr(case["c"]))
        assert (h == conv.fr(case["h"])).all()
        dom.free()


@pytest.mark.parametrize("log_n,m", [(1, 2), (2, 3), (3, 5), (10, 1000), (14, 9000), (16, (1 << 16) - 3), (18, 1 << 18), (20, 1 << 20), (21, (1 << 21) - 77)])
def test_compute_h_vs_oracle(ctx, orc, log_n, m):
    dom = ctx.domain(log_n)
    a, b, c = orc.scalars(3, 0, m), orc.scalars(4, 0, m), orc.scalars(5, 0, m)
    assert (ctx.compute_h(dom, a, b, c) == orc.compute_h(a, b, c, log_n)).all()
    dom.free()


def test_compute_h_rejects_oversized_input(ctx, orc, pkg):
    dom = ctx.domain(4)
    a = orc.scalars(3, 0, 17)
    with pytest.raises(pkg.DvnError):
        ctx.compute_h(dom, a, a, a)
    dom.free()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
@pytest.mark.parametrize("n,batch", [(3000, 5), (1, 3), (70000, 2), (257, 16)])
def test_msm_g1_batched_equals_individual_msms(ctx, orc, n, batch):
    """dvn_msm_g1_batched: B scalar vectors over one point set in one pass == B separate MultiExps (oracle), incl. an all-zero
    vector, a vector of ones (one hot bucket per window) and r - 1 scalars."""
    pts_host = orc.g1_synthetic_points(0xBA7C4, n)
    pts = ctx.upload_g1(pts_host)
    vecs = [orc.scalars(50 + b, b % 2, n) for b in range(batch)]
    vecs[0] = np.zeros((n, 4), dtype=np.uint64)
    one = orc.fr_to_mont(np.array([1, 0, 0, 0], dtype=np.uint64))
    if batch > 2:
        vecs[2] = np.tile(one, (n, 1))
    rm1 = np.array([0x43E1F593F0000000, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029], dtype=np.uint64)   # r - 1
    vecs[1][0] = orc.fr_to_mont(rm1)
    got = ctx.msm_g1_batched(pts, np.stack(vecs))
    for b in range(batch):
        assert (orc.g1_jac_to_affine(got[b]) == orc.msm_g1(pts_host, vecs[b])).all(), b
    assert (got[0][8:12] == 0).all()                      # infinity: Z = 0
    pts.free()


@pytest.mark.gpu
def test_msm_g1_batched_argument_errors(ctx, pkg, orc):
    pts = ctx.upload_g1(orc.g1_synthetic_points(1, 64))
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1_batched(pts, np.zeros((2, 65, 4), dtype=np.uint64))       # len(points) != len(scalars)
    assert ctx.msm_g1_batched(pts, np.zeros((0, 64, 4), dtype=np.uint64)).shape == (0, 12)
    pts.free()


@pytest.mark.gpu
def test_outstanding_tickets_make_synchronous_calls_fail_cleanly(ctx, orc, pkg):
    """The four lanes are shared by tickets and synchronous calls: with four tickets outstanding an MSM and a proof fail with
    the documented message (no wait, no partial state, no leaked lane); once the tickets are collected both work again and
    the collected results are right."""
    n = 5000
    ph = orc.g1_synthetic_points(77, n); pts = ctx.upload_g1(ph)
    sc = orc.scalars(8, 0, n); want = orc.msm_g1(ph, sc)
    cs = pkg.R1CS(7, 2000, 9); key = pkg.Groth16Key(ctx, cs, 3); w = cs.witness()
    r = np.array([5, 0, 0, 0], dtype=np.uint64); s = np.array([7, 0, 0, 0], dtype=np.uint64)
    proof = key.prove(w, r, s)
    for _ in range(2):
        tickets = [ctx.msm_g1_submit(pts, sc) for _ in range(4)]
        with pytest.raises(pkg.DvnError, match="too many MSMs in flight"):
            ctx.msm_g1_submit(pts, sc)
        with pytest.raises(pkg.DvnError, match="too many MSMs in flight"):
            ctx.msm_g1(pts, sc)
        with pytest.raises(pkg.DvnError, match="too many MSMs in flight"):
            key.prove(w, r, s)
        first = ctx.msm_g1_collect(tickets.pop(0))                 # one lane back: a single MSM fits, a proof (several lanes) may not
        assert (aff1(orc, first) == want).all()
        assert (aff1(orc, ctx.msm_g1(pts, sc)) == want).all()
        for t in tickets:
            assert (aff1(orc, ctx.msm_g1_collect(t)) == want).all()
        again = key.prove(w, r, s)
        assert all((again[k] == proof[k]).all() for k in proof)
    key.free(); cs.free(); pts.free()


@pytest.mark.gpu
def test_host_scalars_through_the_staged_upload(ctx, orc):
    """Scalar vectors of 16 MiB and more leave host memory through the ring of pinned chunks (four filler threads, own copy
    stream): sizes just below and above the threshold, not a multiple of the chunk, several calls in a row (the ring is reused,
    also while its last chunks of the call before may still be in flight) -- always the result of the device-resident vector."""
    n_max = (1 << 20) + 12345                                             # 32 MiB + a ragged tail
    pts = ctx.synthetic_g1(0xDA71C1, n_max)
    d = ctx.synthetic_scalars(21, 0, n_max)
    host = np.empty((n_max, 4), dtype=np.uint64); ctx.d2h(host, d)
    for n in ((1 << 19) - 1, (1 << 19) + 3, n_max, (1 << 19) + 3, n_max):
        want = ctx.msm_g1(pts, d, n=n)
        got = ctx.msm_g1(pts, np.ascontiguousarray(host[:n]))
        assert (got == want).all(), n
    t = [ctx.msm_g1_submit(pts, np.ascontiguousarray(host[:n_max])) for _ in range(3)]     # three uploads back to back, three lanes
    want = ctx.msm_g1(pts, d, n=n_max)
    for k in t:
        assert (ctx.msm_g1_collect(k) == want).all()
    ctx.free(d); pts.free()
