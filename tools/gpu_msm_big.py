import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
for log_n in (22, 24):
    n = 1 << log_n
    pts = ctx.synthetic_g1(0xDA71C1, n)
    d = ctx.synthetic_scalars(1, 0, n)
    for rep in range(3):
        t = time.time(); r = ctx.msm_g1(pts, d, n=n); dt = time.time() - t
    print(f"msm_g1 2^{log_n}: wall {dt*1e3:.2f} ms device {ctx.last_kernel_ms(0):.2f} accumulate {ctx.last_kernel_ms(1):.2f} -> {ctx.last_kernel_ms(1)*1e6/(n*16):.4f} ns/add", flush=True)
    ctx.free(d); pts.free()
ctx.close()
