import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
n = 1 << 22
pts = ctx.synthetic_g1(0xDA71C1, n)
d = ctx.synthetic_scalars(1, 0, n)
for rep in range(4):
    t = time.time(); r = ctx.msm_g1(pts, d, n=n); dt = time.time() - t
print(f"SEG={os.environ.get('DVN_MSM_SEG')} EXPAND={os.environ.get('DVN_MSM_EXPAND')}: device {ctx.last_kernel_ms(0):.2f} ms accumulate {ctx.last_kernel_ms(1):.2f} ms", flush=True)
