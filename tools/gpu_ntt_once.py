"""Two 2^24 NTTs (forward DIF, inverse DIT) for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
dom = ctx.domain(24); dx = ctx.synthetic_scalars(2, 0, 1 << 24)
for _ in range(2):
    ctx.ntt(dom, dx, False, pkg.DIF)
    ctx.ntt(dom, dx, True, pkg.DIT)
print("ntt ms", ctx.last_kernel_ms(0), flush=True)
ctx.free(dx); dom.free(); ctx.close()
