import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 17) + 3
grp = sys.argv[2] if len(sys.argv) > 2 else "g2"
c = int(sys.argv[3]) if len(sys.argv) > 3 else 18
pts = ctx.synthetic_g2(5, n) if grp == "g2" else ctx.synthetic_g1(5, n)
if grp == "g1":
    pts.set_expanded(c)
msm = ctx.msm_g2 if grp == "g2" else ctx.msm_g1
for kind in (0, 1):
    d = ctx.synthetic_scalars(9, kind, n)
    ref = None; bad = 0
    for rep in range(12):
        r = msm(pts, d, n=n)
        if ref is None: ref = r
        elif not (r == ref).all(): bad += 1
    print(grp, "n", n, "kind", kind, "mismatching repeats:", bad, flush=True)
    ctx.free(d)
