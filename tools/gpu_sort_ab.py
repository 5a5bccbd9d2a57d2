import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    ctx = pkg.Context(0)
    for log_n in (16, 20, 22):
        n = 1 << log_n
        pts = ctx.synthetic_g1(0xDA71C1, n)
        for kind in (0, 1):
            d = ctx.synthetic_scalars(1, kind, n)
            best = 1e9
            for rep in range(4):
                t = time.time(); r = ctx.msm_g1(pts, d, n=n); best = min(best, time.time() - t)
            print(f"  2^{log_n} kind {kind}: wall {best*1e3:.3f} ms device {ctx.last_kernel_ms(0):.3f} accumulate {ctx.last_kernel_ms(1):.3f} res {int(r[0]) & 0xffffffff:08x}", flush=True)
            ctx.free(d)
        pts.free()
else:
    for mode in ("tile", "staged", "tile", "staged"):
        env = dict(os.environ); env["DVN_MSM_SORT"] = mode
        print("sort =", mode, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
