import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0); bls = pkg.Bls12381(ctx)
for log_n in (12, 16, 20, 22):          # 2^12 = one EIP-4844 blob commitment
    n = 1 << log_n
    pts = bls.synthetic_g1(0xDA71C1, n); d = bls.synthetic_scalars(1, 0, n)
    best = 1e9
    for rep in range(5):
        t = time.time(); r = bls.msm_g1(pts, d, n=n); best = min(best, time.time() - t)
    print(f"bls12-381 msm_g1 2^{log_n}: wall {best*1e3:.3f} ms device {ctx.last_kernel_ms(0):.3f} accumulate {ctx.last_kernel_ms(1):.3f} -> {n/best/1e6:.1f} Mscalar/s", flush=True)
    ctx.free(d); pts.free()
for log_n in (12, 20, 24):
    n = 1 << log_n; dom = bls.domain(log_n); dx = bls.synthetic_scalars(2, 0, n)
    for rep in range(3):
        bls.ntt(dom, dx, False, pkg.DIF); f = ctx.last_kernel_ms(0)
        bls.ntt(dom, dx, True, pkg.DIT); i = ctx.last_kernel_ms(0)
    print(f"bls12-381 ntt 2^{log_n}: fwd {f:.3f} ms inv {i:.3f} ms", flush=True)
    ctx.free(dx); dom.free()
