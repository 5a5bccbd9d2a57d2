"""NTT 2^20 / 2^24 timing probe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
for ln in (20, 24):
    dom = ctx.domain(ln); dx = ctx.synthetic_scalars(2, 0, 1 << ln)
    f, i, c = [], [], []
    for _ in range(5):
        ctx.ntt(dom, dx, False, pkg.DIF); f.append(ctx.last_kernel_ms(0))
        ctx.ntt(dom, dx, True, pkg.DIT); i.append(ctx.last_kernel_ms(0))
        ctx.ntt(dom, dx, False, pkg.DIT, True); c.append(ctx.last_kernel_ms(0))
        ctx.ntt(dom, dx, True, pkg.DIF, True)
    print(f"2^{ln}: fwd DIF {np.median(f):.3f} ms, inv DIT {np.median(i):.3f} ms, coset fwd DIT {np.median(c):.3f} ms", flush=True)
    ctx.free(dx); dom.free()
ctx.close()
