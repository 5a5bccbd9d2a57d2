"""Timing of the NTT on one GPU: median kernel ms per 2^log_n transform for the transform kinds computeH uses (each followed
by its exact inverse), and a check that the sequence returns the input vector.  Usage: python tools/gpu_ntt_ab.py 24 22 20"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
ctx = pkg.Context(0)
for ln in (int(a) for a in (sys.argv[1:] or ["24", "20"])):
    dom = ctx.domain(ln)
    dx = ctx.synthetic_scalars(2, 0, 1 << ln)
    x0 = np.empty((1 << ln, 4), dtype=np.uint64); ctx.d2h(x0, dx)
    res = {}
    for name, (inv, dec, coset), (binv, bdec, bcoset) in (("fwd_dif", (False, pkg.DIF, False), (True, pkg.DIT, False)),
                                                         ("inv_dif", (True, pkg.DIF, False), (False, pkg.DIT, False)),
                                                         ("fwd_dit_coset", (False, pkg.DIT, True), (True, pkg.DIF, True))):
        a, b = [], []
        for _ in range(7):
            ctx.ntt(dom, dx, inv, dec, coset); a.append(ctx.last_kernel_ms(0))
            ctx.ntt(dom, dx, binv, bdec, bcoset); b.append(ctx.last_kernel_ms(0))
        res[name] = round(float(np.median(a)), 4); res[name + "_back"] = round(float(np.median(b)), 4)
    x1 = np.empty_like(x0); ctx.d2h(x1, dx)
    if ln <= 24:
        da, db, dc, dh = (ctx.synthetic_scalars(10 + s, 0, 1 << ln) for s in (1, 2, 3, 4))
        hs = []
        for _ in range(5):
            ctx.compute_h_device(dom, da, db, dc, 1 << ln, dh); hs.append(ctx.last_kernel_ms(0))
        res["compute_h"] = round(float(np.median(hs)), 4)
        for d in (da, db, dc, dh): ctx.free(d)
    print(f"log_n={ln} round_trip={bool((x0 == x1).all())} ms={res}", flush=True)
    ctx.free(dx); dom.free()
ctx.close()
