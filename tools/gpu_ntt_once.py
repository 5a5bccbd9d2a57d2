"""2^24 NTTs (forward DIF, inverse DIT; DVN_NTT_ONCE_ROUNDS pairs, default 8) for rocprofv3 kernel-trace and --pmc passes.
Several rounds so that the per-kernel AVERAGE is the steady state: the first transforms of a process also pay the first touch
of the 512 MiB vector and of the twiddle tables."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
dom = ctx.domain(24); dx = ctx.synthetic_scalars(2, 0, 1 << 24)
for _ in range(int(os.environ.get("DVN_NTT_ONCE_ROUNDS", "8"))):
    ctx.ntt(dom, dx, False, pkg.DIF)
    ctx.ntt(dom, dx, True, pkg.DIT)
print("ntt ms", ctx.last_kernel_ms(0), flush=True)
ctx.free(dx); dom.free(); ctx.close()
