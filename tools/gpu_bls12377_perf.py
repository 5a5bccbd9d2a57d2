import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0); bls = pkg.Bls12377(ctx)
for log_n in (18, 20, 22):
    n = 1 << log_n
    pts = bls.synthetic_g1(0xDA71C1, n); d = bls.synthetic_scalars(1, 0, n)
    for rep in range(3):
        t = time.time(); r = bls.msm_g1(pts, d, n=n); dt = time.time() - t
    print(f"bls12-377 msm_g1 2^{log_n}: wall {dt*1e3:.2f} ms device {ctx.last_kernel_ms(0):.2f} accumulate {ctx.last_kernel_ms(1):.2f} -> {n/dt/1e6:.1f} Mscalar/s", flush=True)
    ctx.free(d); pts.free()
for log_n in (20, 24):
    n = 1 << log_n; dom = bls.domain(log_n); dx = bls.synthetic_scalars(2, 0, n)
    for rep in range(3):
        bls.ntt(dom, dx, False, pkg.DIF); f = ctx.last_kernel_ms(0)
        bls.ntt(dom, dx, True, pkg.DIT); i = ctx.last_kernel_ms(0)
    print(f"bls12-377 ntt 2^{log_n}: fwd {f:.3f} ms inv {i:.3f} ms", flush=True)
    ctx.free(dx); dom.free()
