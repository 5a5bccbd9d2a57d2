import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1 << log_n
pts = ctx.synthetic_g2(5, n); d = ctx.synthetic_scalars(1, kind, n)
for rep in range(3):
    t = time.time(); r = ctx.msm_g2(pts, d, n=n); dt = time.time() - t
print(f"G2 2^{log_n} kind {kind}: wall {dt*1e3:.3f} ms device {ctx.last_kernel_ms(0):.3f} accumulate {ctx.last_kernel_ms(1):.3f}", flush=True)
