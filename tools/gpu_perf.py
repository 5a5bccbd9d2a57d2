"""Quick perf + parity probe used while optimising (not part of the test suite)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc
ctx = pkg.Context(0)
what = sys.argv[1:] or ["msm", "ntt"]
if "msm" in what:
    for n in (1000, 1 << 14, 1 << 17):
        ph = orc.g1_synthetic_points(0xDA71C1, n); pts = ctx.upload_g1(ph)
        for kind in (0, 1):
            sc = orc.scalars(1 + n, kind, n)
            ok = (orc.g1_jac_to_affine(ctx.msm_g1(pts, sc)) == orc.msm_g1(ph, sc)).all()
            print(f"parity msm_g1 n={n} kind={kind}: {'PASS' if ok else 'FAIL'}", flush=True)
        pts.free()
    one = orc.scalars(1, 0, 1)
    n = 1 << 15; ph = orc.g1_synthetic_points(21, n); pts = ctx.upload_g1(ph); sc = np.repeat(one, n, axis=0)
    print("parity hot bucket:", "PASS" if (orc.g1_jac_to_affine(ctx.msm_g1(pts, sc)) == orc.msm_g1(ph, sc)).all() else "FAIL", flush=True)
    pts.free()
    for log_n in (20, 22):
        n = 1 << log_n
        pts = ctx.synthetic_g1(0xDA71C1, n)
        for kind in (0, 1):
            d = ctx.synthetic_scalars(1, kind, n)
            for rep in range(3):
                t = time.time(); r = ctx.msm_g1(pts, d, n=n); dt = time.time() - t
            print(f"msm_g1 2^{log_n} kind={kind}: wall {dt*1e3:.2f} ms device {ctx.last_kernel_ms(0):.2f} ms accumulate {ctx.last_kernel_ms(1):.2f} ms -> {n/dt/1e6:.1f} Mscalar/s", flush=True)
            ctx.free(d)
        pts.free()
if "g2" in what:
    for n in (1000, 1 << 13):
        ph = orc.g2_synthetic_points(5, n); pts = ctx.upload_g2(ph); sc = orc.scalars(9, 0, n)
        ok = (orc.g2_jac_to_affine(ctx.msm_g2(pts, sc)) == orc.msm_g2(ph, sc)).all()
        print(f"parity msm_g2 n={n}: {'PASS' if ok else 'FAIL'}", flush=True); pts.free()
    for log_n in (18, 20):
        n = 1 << log_n
        pts = ctx.synthetic_g2(5, n); d = ctx.synthetic_scalars(1, 0, n)
        for rep in range(2):
            t = time.time(); r = ctx.msm_g2(pts, d, n=n); dt = time.time() - t
        print(f"msm_g2 2^{log_n}: wall {dt*1e3:.2f} ms device {ctx.last_kernel_ms(0):.2f} accumulate {ctx.last_kernel_ms(1):.2f}", flush=True)
        ctx.free(d); pts.free()
if "ntt" in what:
    for log_n in (12, 16):
        n = 1 << log_n; dom = ctx.domain(log_n); x = orc.scalars(2, 0, n)
        for inv, dec, cos in ((0, 0, 0), (1, 1, 0), (0, 1, 1), (1, 0, 1)):
            y = x.copy(); ctx.ntt(dom, y, bool(inv), dec, bool(cos))
            print(f"parity ntt 2^{log_n} {inv}{dec}{cos}:", "PASS" if (y == orc.ntt(x, bool(inv), dec, bool(cos))).all() else "FAIL", flush=True)
        dom.free()
    for log_n in (20, 22, 24):
        n = 1 << log_n; dom = ctx.domain(log_n)
        dx = ctx.synthetic_scalars(2, 0, n)
        for rep in range(3):
            ctx.ntt(dom, dx, False, pkg.DIF); f = ctx.last_kernel_ms(0)
            ctx.ntt(dom, dx, True, pkg.DIT); i = ctx.last_kernel_ms(0)
        ctx.ntt(dom, dx, False, pkg.DIT, True); fc = ctx.last_kernel_ms(0)
        ctx.ntt(dom, dx, True, pkg.DIF, True); ic = ctx.last_kernel_ms(0)
        print(f"ntt 2^{log_n}: fwd {f:.3f} ms inv {i:.3f} ms coset fwd(DIT) {fc:.3f} inv(DIF) {ic:.3f}; fwd+inv {n*2/(f+i)/1e6:.1f} Melem/ms", flush=True)
        ctx.free(dx); dom.free()
ctx.close()
