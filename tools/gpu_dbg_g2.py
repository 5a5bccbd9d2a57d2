import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc
orc.build()
ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 17) + 3
ph = orc.g2_synthetic_points(5, n)
pts = ctx.upload_g2(ph)
for kind in (0, 1):
    sc = orc.scalars(9, kind, n)
    got = orc.g2_jac_to_affine(ctx.msm_g2(pts, sc)); want = orc.msm_g2(ph, sc)
    print("kind", kind, "ok" if (got == want).all() else "MISMATCH", flush=True)
