import os, sys, time, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc
ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print("gen", flush=True)
ph = orc.g2_synthetic_points(5, n)
print("upload", flush=True)
pts = ctx.upload_g2(ph)
print("download", flush=True)
print((pts.download() == ph).all(), flush=True)
sc = orc.scalars(9, 0, n)
print("msm", flush=True)
t = time.time(); r = ctx.msm_g2(pts, sc); print("msm done", time.time() - t, ctx.last_kernel_ms(0), ctx.last_kernel_ms(1), flush=True)
print((orc.g2_jac_to_affine(r) == orc.msm_g2(ph, sc)).all())
