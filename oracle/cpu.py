"""ctypes binding of the C oracle (oracle/liboracle_bn254.so).  TEST INFRASTRUCTURE ONLY -- see bn254_cpu.c."""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_bn254.so")
_U64P = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("bn254_cpu.c", "bn254_field.h", "bn254_ec_tmpl.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle_bn254.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))   # before libgomp starts: not one thread per visible CPU
        _lib = C.CDLL(_SO)
        _lib.orc_ntt_roundtrip_seconds.restype = C.c_double
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_U64P)


def u64(shape):
    return np.zeros(shape, dtype=np.uint64)


def scalars(seed: int, kind: int, n: int) -> np.ndarray:
    out = u64((n, 4)); lib().orc_scalars(C.c_uint64(seed), kind, C.c_size_t(n), _p(out)); return out


def _bin(name, a, b, width):
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    r = u64(width); getattr(lib(), name)(_p(a), _p(b), _p(r)); return r


def fp_mul(a, b): return _bin("orc_fp_mul", a, b, 4)
def fr_mul(a, b): return _bin("orc_fr_mul", a, b, 4)
def fp_add(a, b): return _bin("orc_fp_add", a, b, 4)
def fp_sub(a, b): return _bin("orc_fp_sub", a, b, 4)
def fr_add(a, b): return _bin("orc_fr_add", a, b, 4)
def fr_sub(a, b): return _bin("orc_fr_sub", a, b, 4)
def fp2_mul(a, b): return _bin("orc_fp2_mul", a, b, 8)


def _un(name, a, width):
    a = np.ascontiguousarray(a, dtype=np.uint64); r = u64(width); getattr(lib(), name)(_p(a), _p(r)); return r


def fp_inv(a): return _un("orc_fp_inv", a, 4)
def fr_inv(a): return _un("orc_fr_inv", a, 4)
def fp2_inv(a): return _un("orc_fp2_inv", a, 8)
def fp_to_mont(a): return _un("orc_fp_to_mont", a, 4)
def fp_from_mont(a): return _un("orc_fp_from_mont", a, 4)
def fr_to_mont(a): return _un("orc_fr_to_mont", a, 4)
def fr_from_mont(a): return _un("orc_fr_from_mont", a, 4)


def fr_vec_from_mont(v: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.uint64); out = np.empty_like(v)
    lib().orc_fr_vec_from_mont(_p(v), _p(out), C.c_size_t(v.shape[0])); return out


def g1_generator(): r = u64(8); lib().orc_g1_generator(_p(r)); return r
def g2_generator(): r = u64(16); lib().orc_g2_generator(_p(r)); return r
def g1_on_curve(p): return bool(lib().orc_g1_on_curve(_p(np.ascontiguousarray(p, dtype=np.uint64))))
def g2_on_curve(p): return bool(lib().orc_g2_on_curve(_p(np.ascontiguousarray(p, dtype=np.uint64))))
def g1_add(a, b): return _bin("orc_g1_add", a, b, 8)
def g2_add(a, b): return _bin("orc_g2_add", a, b, 16)


def _int_to_limbs(k: int) -> np.ndarray:
    return np.array([(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def g1_mul(p, k: int): return _bin("orc_g1_mul", p, _int_to_limbs(k), 8)
def g2_mul(p, k: int): return _bin("orc_g2_mul", p, _int_to_limbs(k), 16)
def g1_xyzz_to_affine(p): return _un("orc_g1_xyzz_to_affine", p, 8)
def g2_xyzz_to_affine(p): return _un("orc_g2_xyzz_to_affine", p, 16)
def g1_jac_to_affine(p): return _un("orc_g1_jac_to_affine", p, 8)
def g2_jac_to_affine(p): return _un("orc_g2_jac_to_affine", p, 16)


def g1_synthetic_points(seed: int, n: int, nthreads: int = 0) -> np.ndarray:
    out = u64((n, 8)); lib().orc_g1_synthetic_points(C.c_uint64(seed), C.c_size_t(n), _p(out), _nt(nthreads)); return out


def g2_synthetic_points(seed: int, n: int, nthreads: int = 0) -> np.ndarray:
    out = u64((n, 16)); lib().orc_g2_synthetic_points(C.c_uint64(seed), C.c_size_t(n), _p(out), _nt(nthreads)); return out


def msm_g1(points: np.ndarray, scalars_mont: np.ndarray, nthreads: int = 0) -> np.ndarray:
    points = np.ascontiguousarray(points, dtype=np.uint64); scalars_mont = np.ascontiguousarray(scalars_mont, dtype=np.uint64)
    out = u64(8)
    rc = lib().orc_msm_g1(_p(points), C.c_size_t(points.shape[0]), _p(scalars_mont), C.c_size_t(scalars_mont.shape[0]), _nt(nthreads), _p(out))
    if rc != 0:
        raise ValueError("len(points) != len(scalars)")
    return out


def msm_g1_window_sums(points: np.ndarray, scalars_mont: np.ndarray, c: int, nthreads: int = 0) -> np.ndarray:
    points = np.ascontiguousarray(points, dtype=np.uint64); scalars_mont = np.ascontiguousarray(scalars_mont, dtype=np.uint64)
    nwin = 254 // c + 1
    out = u64((nwin, 16))
    lib().orc_msm_g1_window_sums(_p(points), _p(scalars_mont), C.c_size_t(points.shape[0]), c, _nt(nthreads), _p(out))
    return out


def msm_g2(points: np.ndarray, scalars_mont: np.ndarray, nthreads: int = 0) -> np.ndarray:
    points = np.ascontiguousarray(points, dtype=np.uint64); scalars_mont = np.ascontiguousarray(scalars_mont, dtype=np.uint64)
    out = u64(16)
    rc = lib().orc_msm_g2(_p(points), C.c_size_t(points.shape[0]), _p(scalars_mont), C.c_size_t(scalars_mont.shape[0]), _nt(nthreads), _p(out))
    if rc != 0:
        raise ValueError("len(points) != len(scalars)")
    return out


def ntt(data: np.ndarray, inverse: bool, decimation: int, coset: bool, nthreads: int = 0) -> np.ndarray:
    """Returns a transformed copy.  data: (n, 4) uint64 Montgomery, n a power of two."""
    out = np.array(data, dtype=np.uint64, copy=True, order="C")
    n = out.shape[0]; log_n = n.bit_length() - 1
    assert 1 << log_n == n
    rc = lib().orc_ntt(_p(out), log_n, int(inverse), int(decimation), int(coset), _nt(nthreads))
    assert rc == 0
    return out


def ntt_roundtrip_seconds(data: np.ndarray, reps: int, nthreads: int = 0) -> float:
    n = data.shape[0]; log_n = n.bit_length() - 1
    return float(lib().orc_ntt_roundtrip_seconds(_p(data), log_n, reps, _nt(nthreads)))


def compute_h(a, b, c, log_n: int, nthreads: int = 0) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64); c = np.ascontiguousarray(c, dtype=np.uint64)
    assert a.shape == b.shape == c.shape
    out = u64((1 << log_n, 4))
    rc = lib().orc_compute_h(_p(a), _p(b), _p(c), C.c_size_t(a.shape[0]), log_n, _p(out), _nt(nthreads))
    if rc != 0:
        raise ValueError("len(a) > domain size")
    return out


def num_threads() -> int:
    return int(lib().orc_num_threads())


def _nt(nthreads: int) -> int:
    """0 = as many threads as this process may really use (not OpenMP's default of every visible CPU)."""
    return nthreads if nthreads > 0 else usable_cores()


def usable_cores() -> int:
    """CPU cores this process can actually use: the scheduler affinity capped by the cgroup CPU quota (a container can see
    256 CPUs and be allowed 16 CPU-seconds per second; more threads than that only adds throttling)."""
    import math, os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, math.ceil(q / per)))
    except Exception:
        pass
    return n


def groth16_synthetic_expected(cs_seed: int, n_constraints: int, n_public: int, setup_seed: int, r_mont, s_mont):
    """Closed-form proof from the trapdoor for the synthetic instance.  Returns (witness (m,4), {'Ar','Bs','Krs'})."""
    m = n_public + 4 + n_constraints
    wit = u64((m, 4)); proof = u64(32)
    r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
    lib().orc_groth16_synthetic_expected(C.c_uint64(cs_seed), C.c_uint32(n_constraints), C.c_uint32(n_public), C.c_uint64(setup_seed),
                                         _p(r_mont), _p(s_mont), _p(wit), _p(proof))
    return wit, {"Ar": proof[0:8].copy(), "Bs": proof[8:24].copy(), "Krs": proof[24:32].copy()}
