import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DVN_MSM_EXPAND"] = "0"
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
log_n = int(sys.argv[1]); c = int(sys.argv[2]); grp = sys.argv[3] if len(sys.argv) > 3 else "g2"
n = 1 << log_n
pts = ctx.synthetic_g2(5, n) if grp == "g2" else ctx.synthetic_g1(5, n)
msm = ctx.msm_g2 if grp == "g2" else ctx.msm_g1
d = ctx.synthetic_scalars(1, 0, n)
ref = msm(pts, d, n=n)
pts.set_expanded(c)
bad = 0
for rep in range(6):
    r = msm(pts, d, n=n)
    bad += int(not (r == ref).all())
print(f"{grp} 2^{log_n} c{c} grid {os.environ.get('DVN_FIXH_GRID', '512')}: mismatches {bad}/6", flush=True)
