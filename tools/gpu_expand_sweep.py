import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DVN_MSM_EXPAND"] = "0"          # plain sets unless set_expanded is called
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
grp = sys.argv[1] if len(sys.argv) > 1 else "g1"
for log_n in ((16, 17, 18, 19, 20, 21, 22) if grp == "g1" else (15, 16, 17, 18, 19, 20, 21)):
    n = 1 << log_n
    pts = ctx.synthetic_g1(0xDA71C1, n) if grp == "g1" else ctx.synthetic_g2(5, n)
    msm = ctx.msm_g1 if grp == "g1" else ctx.msm_g2
    d = ctx.synthetic_scalars(1, 0, n)
    row = []
    for c in (0, 14, 15, 16, 17, 18, 19, 20):
        if c and c > log_n + 1: continue
        pts.set_expanded(c)
        best = 1e9
        for rep in range(4):
            t = time.time(); r = msm(pts, d, n=n); best = min(best, time.time() - t)
        row.append(f"c{c}:{best*1e3:.2f}")
    print(f"{grp} 2^{log_n}: " + "  ".join(row), flush=True)
    ctx.free(d); pts.free()
