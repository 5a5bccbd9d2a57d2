"""The C++ mirror of the reference's `prover` package: UseGPUProver gate, SetProver seam,
GPU -> CPU fallback with the reference's message, per-curve dispatch.  The 'CPU prover' registered here is a test
double built on the oracle (in the node it is gnark's groth16.Prove)."""
import ctypes as C
import os

import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def mont(o, x):
    return np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)


@pytest.fixture()
def L(pkg):
    lib = pkg.load_library()
    yield lib
    lib.dvn_prover_set_prover(None, None)
    lib.dvn_prover_set_cpu_backend(None, None)
    lib.dvn_prover_set_use_gpu(0)


def test_use_gpu_prover_env_gate(L, monkeypatch):
    # prover/config.go:18-26: GPU_PROVER in {true, y, 1, yes}
    for v, want in (("true", 1), ("y", 1), ("1", 1), ("yes", 1), ("no", 0), ("TRUE", 0), ("", 0)):
        monkeypatch.setenv("GPU_PROVER", v)
        L.dvn_prover_reload_env()
        assert L.dvn_prover_use_gpu() == want, v
    monkeypatch.delenv("GPU_PROVER")
    L.dvn_prover_reload_env()
    assert L.dvn_prover_use_gpu() == 0


def test_cpu_path_without_backend_errors_loudly(L, pkg):
    """UseGPUProver = false -> CPUProver; with no CPU backend registered the call FAILS (no silent CPU compute)."""
    cs = pkg.R1CS(11, 13, 3)
    out = np.zeros(32, dtype=np.uint64); r = np.zeros(4, dtype=np.uint64)
    w = cs.witness()
    rc = L.dvn_prover_prove(None, 1, cs.h, None, w.ctypes.data, r.ctypes.data, r.ctypes.data, out.ctypes.data)
    assert rc == -5 and b"CPU prover not linked" in L.dvn_last_error(None)
    assert L.dvn_prover_last_backend() == 2
    cs.free()


def test_gpu_prover_without_a_device_context_is_refused(L, pkg):
    """prover/prover_cpu_test.go:7-25: in a build without GPU support GPUProver and GPUProverWithWitness panic with
    "GPU prover not supported in this build".  This library's counterpart of that build is a process without a device
    context; C has no panic, the call fails with the same text -- also through Prove when GPU_PROVER is on and no CPU
    backend is registered (the fallback then fails loudly too)."""
    cs = pkg.R1CS(11, 13, 3)
    out = np.zeros(32, dtype=np.uint64); r = np.zeros(4, dtype=np.uint64)
    w = cs.witness()
    rc = L.dvn_prover_gpu_prover(None, 1, cs.h, None, w.ctypes.data, r.ctypes.data, r.ctypes.data, out.ctypes.data)
    assert rc == -5 and L.dvn_last_error(None) == b"GPU prover not supported in this build"
    assert L.dvn_prover_last_backend() == 1
    assert L.dvn_prover_gpu_prover(None, 0, None, None, None, None, None, None) == -5        # GPUProver(0, nil, nil, nil)
    rc = L.dvn_prover_cpu_prover(None, 1, cs.h, None, w.ctypes.data, r.ctypes.data, r.ctypes.data, out.ctypes.data)
    assert rc == -5 and b"CPU prover not linked" in L.dvn_last_error(None)
    cs.free()


def _oracle_backend(orc, seeds, calls):
    from davinci_node_amd import binding
    L_ = binding.load_library()
    def cb(curve, ccs, pk, wit, r, s, proof, user):
        # the reference's CPUProverWithWitness gets (ccs, pk, witness) (prover_gpu.go:126-131): so does the registered backend
        assert ccs, "the CPU backend receives the constraint system"
        info = [C.c_uint32(), C.c_uint32(), C.c_uint32()]
        assert L_.dvn_r1cs_info(C.c_void_p(ccs), *[C.byref(x) for x in info]) == 0 and info[2].value == seeds[1]
        calls.append(curve)
        rr = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_uint64)), (4,)).copy()
        ss = np.ctypeslib.as_array(C.cast(s, C.POINTER(C.c_uint64)), (4,)).copy()
        _, p = orc.groth16_synthetic_expected(*seeds, rr, ss)
        dst = np.ctypeslib.as_array(C.cast(proof, C.POINTER(C.c_uint64)), (32,))
        dst[0:8] = p["Ar"]; dst[8:24] = p["Bs"]; dst[24:32] = p["Krs"]
        return 0
    return binding.CPU_BACKEND_T(cb)


def test_set_prover_seam_and_cpu_backend(L, pkg, opy, orc):
    from davinci_node_amd import binding
    cs = pkg.R1CS(7, 100, 3)
    calls = []
    be = _oracle_backend(orc, (7, 100, 3, 3), calls)
    L.dvn_prover_set_cpu_backend(C.cast(be, C.c_void_p), None)
    out = np.zeros(32, dtype=np.uint64); w = cs.witness()
    r, s = mont(opy, 5), mont(opy, 6)
    assert L.dvn_prover_prove(None, 1, cs.h, None, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) == 0
    assert calls == [1] and L.dvn_prover_last_backend() == 2
    _, want = orc.groth16_synthetic_expected(7, 100, 3, 3, r, s)
    assert (out[0:8] == want["Ar"]).all() and (out[24:32] == want["Krs"]).all()
    # SetProver: tests swap the prover func (tests/csp_census_test.go:47 style)
    seen = []
    def custom(req, user):
        seen.append(1); return 0
    cp = binding.PROVER_FUNC_T(custom)
    L.dvn_prover_set_prover(C.cast(cp, C.c_void_p), None)
    assert L.dvn_prover_prove(None, 1, cs.h, None, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) == 0
    assert seen == [1] and L.dvn_prover_last_backend() == 3 and calls == [1]
    cs.free()


def test_gpu_error_falls_back_to_cpu_with_reference_message(L, pkg, opy, orc, capfd):
    """prover_gpu.go:73-80: any GPU error -> 'GPU proving failed (...), falling back to CPU' -> CPU result.
    No GPU / NULL key makes the GPU leg fail here; the gpu suite repeats this with DVN_FORCE_FAIL on a live device."""
    cs = pkg.R1CS(7, 100, 3)
    calls = []
    be = _oracle_backend(orc, (7, 100, 3, 3), calls)
    L.dvn_prover_set_cpu_backend(C.cast(be, C.c_void_p), None)
    L.dvn_prover_set_use_gpu(1)
    out = np.zeros(32, dtype=np.uint64); w = cs.witness(); r, s = mont(opy, 5), mont(opy, 6)
    assert L.dvn_prover_prove(None, 1, cs.h, None, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) == 0
    assert "GPU proving failed (" in capfd.readouterr().out
    assert calls == [1] and L.dvn_prover_last_backend() == 2
    # a curve without a GPU leg -> same fallback (prover_gpu.go:58-60); all four curves of the reference have one here
    assert L.dvn_prover_prove(None, 9, cs.h, None, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) == 0
    assert "GPU proving not supported for curve unknown" in capfd.readouterr().out
    assert L.dvn_prover_prove(None, 4, cs.h, None, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) == 0
    assert "proving key type mismatch for BW6_761" in capfd.readouterr().out
    assert calls == [1, 9, 4]
    cs.free()


@pytest.mark.gpu
def test_gpu_prover_and_forced_failure_fallback(L, ctx, pkg, opy, orc, capfd):
    cs = pkg.R1CS(7, 1000, 9)
    pk = C.c_void_p()
    assert L.dvn_prover_setup(ctx.h, cs.h, 3, C.byref(pk)) == 0
    calls = []
    be = _oracle_backend(orc, (7, 1000, 9, 3), calls)
    L.dvn_prover_set_cpu_backend(C.cast(be, C.c_void_p), None)
    L.dvn_prover_set_use_gpu(1)
    out = np.zeros(32, dtype=np.uint64); w = cs.witness(); r, s = mont(opy, 123), mont(opy, 456)
    assert L.dvn_prover_prove(ctx.h, 1, cs.h, pk, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) == 0
    assert L.dvn_prover_last_backend() == 1 and calls == []            # GPU leg served it
    _, want = orc.groth16_synthetic_expected(7, 1000, 9, 3, r, s)
    assert (out[0:8] == want["Ar"]).all() and (out[8:24] == want["Bs"]).all() and (out[24:32] == want["Krs"]).all()
    # fault injection: a context created under DVN_FORCE_FAIL fails every device entry point
    os.environ["DVN_FORCE_FAIL"] = "1"
    try:
        bad = pkg.Context(0)
    finally:
        del os.environ["DVN_FORCE_FAIL"]
    out2 = np.zeros(32, dtype=np.uint64)
    assert L.dvn_prover_prove(bad.h, 1, cs.h, pk, w.ctypes.data, r.ctypes.data, s.ctypes.data, out2.ctypes.data) == 0
    assert "GPU proving failed (DVN_FORCE_FAIL" in capfd.readouterr().out
    assert calls == [1] and (out2 == out).all()                       # CPU backend result equals the GPU proof
    bad.close()
    L.dvn_groth16_pk_free(ctx.h, pk); cs.free()


@pytest.mark.gpu
def test_prover_mirror_dispatches_bls12377_to_the_gpu(L, ctx, pkg, capfd):
    """prover_gpu.go:36-43: ecc.BLS12_377 has a GPU leg.  Setup picks the curve from the constraint system, Prove runs on the
    device (backend 1) and yields the trapdoor proof; a key of the wrong curve is an error -> CPU fallback message."""
    from oracle import curves_py as cp
    S = cp.BLS12_377
    cs = pkg.R1CS(3, 200, 2, curve="bls12377")
    assert L.dvn_r1cs_curve(cs.h) == 2
    pk = C.c_void_p()
    assert L.dvn_prover_setup(ctx.h, cs.h, 9, C.byref(pk)) == 0
    L.dvn_prover_set_prover(None, None)
    L.dvn_prover_set_use_gpu(1)
    out = np.zeros(48, dtype=np.uint64); w = cs.witness()
    r = np.array(S.fr_to_limbs(5), dtype=np.uint64); s = np.array(S.fr_to_limbs(11), dtype=np.uint64)
    assert L.dvn_prover_prove(ctx.h, 2, cs.h, pk, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) == 0
    assert L.dvn_prover_last_backend() == 1
    want = cp.groth16_expected_from_trapdoor_on(cp.SyntheticR1CSOn(S, 200, 2, 3), 9, 5, 11)
    assert list(out[0:12]) == S.g1_to_limbs(want["Ar"]) and list(out[12:36]) == S.g2_to_limbs(want["Bs"]) and list(out[36:48]) == S.g1_to_limbs(want["Krs"])
    # the BLS12-377 key handed to the BN254 leg: type mismatch -> fallback (no CPU backend registered -> error code)
    L.dvn_prover_set_cpu_backend(None, None)
    assert L.dvn_prover_prove(ctx.h, 1, cs.h, pk, w.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data) != 0
    assert "GPU proving failed (" in capfd.readouterr().out
    L.dvn_prover_set_use_gpu(0)
    assert L.dvn_groth16_pk_free(ctx.h, pk) == 0
    cs.free()
