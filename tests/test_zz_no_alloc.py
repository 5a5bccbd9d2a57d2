"""dvn_msm_g1_static -- the entry point that allocates nothing -- against dvn_msm_g1 through the C-ABI.

Same kernels over the library's fixed scratch array instead of the hipMalloc'ed workspace: the results must agree bit for
bit, on two calls in a row (the second smaller, so stale contents of the shared array are in play), and the oracle is
consulted on the small case.  Also checked: the device's free memory does not move across the calls, sizes beyond the
maximum and host scalars are rejected with a clear message.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes(pkg):
    """hipMemGetInfo of the current device from the HIP runtime libdvn_prover.so itself is bound to: dlsym on the library's
    handle searches its dependencies, so this never picks up a second runtime another package may have loaded (torch ships one)."""
    fn = pkg.load_library().hipMemGetInfo
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    free, total = C.c_size_t(), C.c_size_t()
    assert fn(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_msm_matches_msm_g1_two_calls(ctx, pkg, orc):
    n1, n2 = 50000, 3001                                  # expanded set (>= 2^15 points), then a plain-set sub-range
    pts = ctx.synthetic_g1(0xDA71C1, n1)
    small = ctx.synthetic_g1(0xBEEF, n2)
    d1 = ctx.synthetic_scalars(11, 0, n1)
    d2 = ctx.synthetic_scalars(12, 1, n2)                 # witness-like mixture: hot buckets
    try:
        want1 = ctx.msm_g1(pts, d1, n=n1)
        want2 = ctx.msm_g1(small, d2, n=n2)
        ctx.synchronize()
        free0 = _free_bytes(pkg)
        got1 = ctx.msm_g1_static(pts, d1, n1)
        got2 = ctx.msm_g1_static(small, d2, n2)           # second call: smaller input, scratch holds the first call's data
        got1b = ctx.msm_g1_static(pts, d1, n1)
        assert _free_bytes(pkg) == free0, "the allocation-free entry point changed the device's free memory"
        assert (got1 == want1).all() and (got1b == want1).all()
        assert (got2 == want2).all()
        # the oracle on the small case (independent of dvn_msm_g1)
        sc = np.zeros((n2, 4), dtype=np.uint64)
        ctx.d2h(sc, d2)
        assert (orc.g1_jac_to_affine(got2) == orc.msm_g1(small.download(), sc)).all()
        # empty input: infinity, like MultiExp on empty slices
        inf = ctx.msm_g1_static(small, d2, 0)
        assert (inf == ctx.msm_g1(small, d2, n=0)).all()
    finally:
        ctx.free(d1); ctx.free(d2); pts.free(); small.free()


def test_static_msm_rejects_what_it_cannot_serve(ctx, pkg):
    n = pkg.binding.MSM_STATIC_MAX_N + 1
    pts = ctx.synthetic_g1(5, n)
    d = ctx.synthetic_scalars(6, 0, n)
    try:
        with pytest.raises(pkg.DvnError) as e:
            ctx.msm_g1_static(pts, d, n)
        assert e.value.code == -1 and "DVN_MSM_STATIC_MAX_N" in str(e.value)
        with pytest.raises(pkg.DvnError) as e:
            ctx.msm_g1_static(pts, d, 16, flags=0)        # host-scalar flag: refused, a host copy could allocate
        assert e.value.code == -1 and "device scalars" in str(e.value)
        # the limit itself is served
        ok = ctx.msm_g1_static(pts, d, n - 1)
        assert (ok == ctx.msm_g1(pts, d, n=n - 1)).all()
    finally:
        ctx.free(d); pts.free()
