import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
c = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pts = ctx.synthetic_g1(0xDA71C1, n); d = ctx.synthetic_scalars(1, 0, n)
for rep in range(3):
    t = time.time(); r0 = ctx.msm_g1(pts, d, n=n); dt0 = time.time() - t
print(f"plain    : wall {dt0*1e3:.2f} ms device {ctx.last_kernel_ms(0):.2f} accumulate {ctx.last_kernel_ms(1):.2f}", flush=True)
if c:
    pts.set_expanded(c)
    for rep in range(3):
        t = time.time(); r1 = ctx.msm_g1(pts, d, n=n); dt1 = time.time() - t
    print(f"expanded{c}: wall {dt1*1e3:.2f} ms device {ctx.last_kernel_ms(0):.2f} accumulate {ctx.last_kernel_ms(1):.2f} same={bool((r0 == r1).all())}", flush=True)
