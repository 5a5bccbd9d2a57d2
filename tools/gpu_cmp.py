import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
for curve in ("g1", "g2"):
    for log_n in ((16, 17, 18, 19, 20, 21, 22) if curve == "g1" else (16, 18, 19, 20, 21)):
        n = 1 << log_n
        pts = ctx.synthetic_g1(0xDA71C1, n) if curve == "g1" else ctx.synthetic_g2(5, n)
        d = ctx.synthetic_scalars(1, 0, n)
        f = ctx.msm_g1 if curve == "g1" else ctx.msm_g2
        for rep in range(4):
            t = time.time(); r = f(pts, d, n=n); dt = time.time() - t
        print(f"EXPAND={os.environ.get('DVN_MSM_EXPAND')} {curve} 2^{log_n}: wall {dt*1e3:.2f} device {ctx.last_kernel_ms(0):.2f} acc {ctx.last_kernel_ms(1):.2f}", flush=True)
        ctx.free(d); pts.free()
