import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
n = 1 << 22
pts = ctx.synthetic_g1(0xDA71C1, n); d = ctx.synthetic_scalars(1, 0, n)
host = np.empty((n, 4), dtype=np.uint64); ctx.d2h(host, d)
for rep in range(3):
    t = time.time(); r0 = ctx.msm_g1(pts, d, n=n); dt_dev = time.time() - t
for rep in range(3):
    t = time.time(); r1 = ctx.msm_g1(pts, host); dt_host = time.time() - t
assert (r0 == r1).all()
print(f"2^22 G1 MSM, points resident: scalars resident {dt_dev*1e3:.2f} ms ({n/dt_dev/1e6:.0f} Mscalar/s); scalars from pageable host memory "
      f"(128 MiB H2D inside the call) {dt_host*1e3:.2f} ms ({n/dt_host/1e6:.0f} Mscalar/s)")
