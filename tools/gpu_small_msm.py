"""Latency of small G1 MSMs (the commitment-hint sizes), one at a time and two in flight.  Usage: python tools/gpu_small_msm.py [log_n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
for ln in (int(a) for a in (sys.argv[1:] or ["10", "12", "14", "15", "16"])):
    n = 1 << ln
    pts = ctx.synthetic_g1(7, n); d = ctx.synthetic_scalars(3, 0, n)
    for _ in range(3): ctx.msm_g1(pts, d, n=n)
    t = time.perf_counter()
    for _ in range(20): ctx.msm_g1(pts, d, n=n)
    one = (time.perf_counter() - t) / 20
    dev = ctx.last_kernel_ms(0)
    t = time.perf_counter()
    for _ in range(20):
        a = ctx.msm_g1_submit(pts, d, n=n); b = ctx.msm_g1_submit(pts, d, n=n)
        ctx.msm_g1_collect(a); ctx.msm_g1_collect(b)
    two = (time.perf_counter() - t) / 20
    print(f"n=2^{ln}: one MSM {one*1e3:.3f} ms wall ({dev:.3f} ms on the device), two in flight {two*1e3:.3f} ms", flush=True)
    want = ctx.msm_g1(pts, d, n=n)
    for c in (13, 15, 16):
        if ln >= 15: break
        pts.set_expanded(c)
        assert (ctx.msm_g1(pts, d, n=n) == want).all()
        for _ in range(3): ctx.msm_g1(pts, d, n=n)
        t = time.perf_counter()
        for _ in range(20): ctx.msm_g1(pts, d, n=n)
        one = (time.perf_counter() - t) / 20
        print(f"   expanded c={c}: one MSM {one*1e3:.3f} ms wall ({ctx.last_kernel_ms(0):.3f} ms on the device)", flush=True)
    pts.set_expanded(0)
    ctx.free(d); pts.free()
ctx.close()
