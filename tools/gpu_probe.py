"""Ad-hoc GPU probe: parity of each primitive vs the oracle at a few sizes + first timings.  (tests/ holds the real suite.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc

ctx = pkg.Context(0)
ok = True
def check(name, cond):
    global ok
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    ok = ok and bool(cond)

# synthetic points on device vs oracle
for n in (1, 5, 100, 1000):
    p = ctx.synthetic_g1(0xDA71C1, n)
    check(f"synthetic_g1 n={n}", (p.download() == orc.g1_synthetic_points(0xDA71C1, n)).all())
    p.free()
p2 = ctx.synthetic_g2(7, 50)
check("synthetic_g2 n=50", (p2.download() == orc.g2_synthetic_points(7, 50)).all())
# upload/download round trip
ph = orc.g1_synthetic_points(3, 300)
pu = ctx.upload_g1(ph)
check("upload/download g1", (pu.download() == ph).all())

for n in (1, 2, 33, 1000, 1 << 12, 1 << 14, 1 << 16):
    ph = orc.g1_synthetic_points(0xDA71C1, n)
    pts = ctx.upload_g1(ph)
    for kind in (0, 1):
        sc = orc.scalars(1 + n, kind, n)
        t = time.time(); got = orc.g1_jac_to_affine(ctx.msm_g1(pts, sc)); dt = time.time() - t
        want = orc.msm_g1(ph, sc)
        check(f"msm_g1 n={n} kind={kind} ({dt*1e3:.1f} ms, dev {ctx.last_kernel_ms(0):.2f} ms, acc {ctx.last_kernel_ms(1):.2f} ms)", (got == want).all())
    pts.free()
for n in (1, 17, 1000, 1 << 12):
    ph = orc.g2_synthetic_points(5, n)
    pts = ctx.upload_g2(ph)
    sc = orc.scalars(9, 0, n)
    t = time.time(); got = orc.g2_jac_to_affine(ctx.msm_g2(pts, sc)); dt = time.time() - t
    check(f"msm_g2 n={n} ({dt*1e3:.1f} ms)", (got == orc.msm_g2(ph, sc)).all())
    pts.free()

for log_n in (1, 2, 3, 5, 8, 9, 11, 12, 16, 18):
    n = 1 << log_n
    dom = ctx.domain(log_n)
    x = orc.scalars(2, 0, n)
    for inv in (0, 1):
        for dec in (0, 1):
            for cos in (0, 1):
                y = x.copy(); ctx.ntt(dom, y, bool(inv), dec, bool(cos))
                check(f"ntt log_n={log_n} inv={inv} dec={dec} coset={cos} ({ctx.last_kernel_ms(0):.3f} ms)", (y == orc.ntt(x, bool(inv), dec, bool(cos))).all())
    if log_n >= 3:
        m = n - 3
        a, b, c = orc.scalars(3, 0, m), orc.scalars(4, 0, m), orc.scalars(5, 0, m)
        check(f"compute_h log_n={log_n}", (ctx.compute_h(dom, a, b, c) == orc.compute_h(a, b, c, log_n)).all())
    dom.free()

# first timings
for log_n in (20, 22):
    n = 1 << log_n
    t = time.time(); pts = ctx.synthetic_g1(0xDA71C1, n); print(f"synthetic_g1 2^{log_n}: {time.time()-t:.2f} s", flush=True)
    sc = orc.scalars(1, 0, n)
    dsc = ctx.to_device(sc)
    for rep in range(3):
        t = time.time(); r = ctx.msm_g1(pts, dsc, n=n); dt = time.time() - t
        print(f"msm_g1 2^{log_n} uniform: wall {dt*1e3:.2f} ms, device {ctx.last_kernel_ms(0):.2f} ms, accumulate {ctx.last_kernel_ms(1):.2f} ms -> {n/dt/1e6:.1f} Mscalar/s", flush=True)
    # closed form check: sum s_i (P0 + i D) = (sum s_i) P0 + (sum i s_i) D
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    can = orc.fr_vec_from_mont(sc)
    vals = [int(can[i,0]) | int(can[i,1])<<64 | int(can[i,2])<<128 | int(can[i,3])<<192 for i in range(n)]
    s0 = sum(vals) % R; s1 = sum(i*v for i, v in enumerate(vals)) % R
    two = pts.download(0, 2)
    P0 = two[0]; negP0 = P0.copy()
    D = None
    # D = P1 - P0 via oracle
    from oracle import bn254_py as o
    p0 = o.g1_from_limbs([int(v) for v in two[0]]); p1 = o.g1_from_limbs([int(v) for v in two[1]])
    d = o.g1_add(p1, o.g1_neg(p0))
    exp = o.g1_add(o.g1_mul(p0, s0), o.g1_mul(d, s1))
    check(f"msm_g1 2^{log_n} closed form", list(orc.g1_jac_to_affine(r)) == o.g1_to_limbs(exp))
    scw = orc.scalars(1, 1, n); ctx.h2d(dsc, scw)
    t = time.time(); r = ctx.msm_g1(pts, dsc, n=n); dt = time.time() - t
    print(f"msm_g1 2^{log_n} witness-like: wall {dt*1e3:.2f} ms, device {ctx.last_kernel_ms(0):.2f} ms, accumulate {ctx.last_kernel_ms(1):.2f} ms", flush=True)
    ctx.free(dsc); pts.free()
for log_n in (20, 24):
    n = 1 << log_n
    dom = ctx.domain(log_n)
    x = orc.scalars(2, 0, n)
    dx = ctx.to_device(x)
    for rep in range(3):
        ctx.ntt(dom, dx, False, pkg.DIF); f = ctx.last_kernel_ms(0)
        ctx.ntt(dom, dx, True, pkg.DIT); i = ctx.last_kernel_ms(0)
        print(f"ntt 2^{log_n}: fwd {f:.3f} ms inv {i:.3f} ms", flush=True)
    y = np.empty_like(x); ctx.d2h(y, dx)
    check(f"ntt 2^{log_n} round trip", (x == y).all())
    ctx.free(dx); dom.free()
print("ALL OK" if ok else "SOME FAILED")
ctx.close()
