"""ctypes binding of include/dvn_prover.h.  Thin: argument marshalling and error mapping only."""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
DIF, DIT = 0, 1
MSM_SCALARS_ON_DEVICE = 1
MSM_STATIC_MAX_N = 65536
NTT_DATA_ON_DEVICE = 1

_U64P = C.POINTER(C.c_uint64)


class DvnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"dvn error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    """The in-tree build; DVN_PROVER_LIB names another build of the SAME library (kernel A/B experiments)."""
    return os.environ.get("DVN_PROVER_LIB") or os.path.join(_CSRC, "libdvn_prover.so")


def build_library(jobs: int = 4) -> str:
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", _CSRC, f"-j{jobs}", "-s", "all"])
    return lib_path()


_lib = None


def load_library():
    """Load libdvn_prover.so.  Fails loudly if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    L = C.CDLL(p)
    vp, sz, u32, i32, u64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32, C.c_uint64
    sig = {
        "dvn_init": (i32, [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
        "dvn_shutdown": (None, [vp]),
        "dvn_last_error": (C.c_char_p, [vp]),
        "dvn_version": (C.c_char_p, []),
        "dvn_test_raise": (i32, [i32]),
        "dvn_bench_mad_rate": (i32, [vp, C.POINTER(C.c_double)]),
        "dvn_malloc": (i32, [vp, sz, C.POINTER(vp)]),
        "dvn_free": (i32, [vp, vp]),
        "dvn_memcpy_h2d": (i32, [vp, vp, vp, sz]),
        "dvn_memcpy_d2h": (i32, [vp, vp, vp, sz]),
        "dvn_synchronize": (i32, [vp]),
        "dvn_points_upload_g1": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "dvn_points_upload_g2": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "dvn_points_synthetic_g1": (i32, [vp, u64, sz, C.POINTER(vp)]),
        "dvn_points_synthetic_g2": (i32, [vp, u64, sz, C.POINTER(vp)]),
        "dvn_scalars_synthetic": (i32, [vp, u64, C.c_int, sz, vp]),
        "dvn_points_set_expanded": (i32, [vp, vp, u32]),
        "dvn_points_download": (i32, [vp, vp, sz, sz, vp]),
        "dvn_points_len": (sz, [vp]),
        "dvn_points_free": (i32, [vp, vp]),
        "dvn_msm_g1": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_msm_g1_static": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_msm_g2": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_msm_g1_batched": (i32, [vp, vp, sz, sz, vp, u32, u32, vp]),
        "dvn_msm_g1_submit": (i32, [vp, vp, sz, sz, vp, u32, C.POINTER(i32)]),
        "dvn_msm_g1_collect": (i32, [vp, i32, vp]),
        "dvn_msm_g2_submit": (i32, [vp, vp, sz, sz, vp, u32, C.POINTER(i32)]),
        "dvn_msm_g2_collect": (i32, [vp, i32, vp]),
        "dvn_msm_plan": (i32, [sz, C.POINTER(u32), C.POINTER(u32)]),
        "dvn_msm_plan_points": (i32, [vp, sz, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "dvn_msm_g1_windows": (i32, [vp, vp, sz, sz, vp, u32, u32, u32, vp]),
        "dvn_msm_g1_windows_submit": (i32, [vp, vp, sz, sz, vp, u32, u32, u32, C.POINTER(i32)]),
        "dvn_msm_g1_windows_collect": (i32, [vp, i32, vp]),
        "dvn_msm_combine_g1": (i32, [vp, vp, u32, u32, vp]),
        "dvn_ntt_domain": (i32, [vp, u32, C.POINTER(vp)]),
        "dvn_ntt_domain_free": (i32, [vp, vp]),
        "dvn_ntt": (i32, [vp, vp, vp, C.c_int, C.c_int, C.c_int, u32]),
        "dvn_groth16_compute_h": (i32, [vp, vp, vp, vp, vp, sz, vp, u32]),
        "dvn_last_kernel_ms": (i32, [vp, C.c_int, C.POINTER(C.c_float)]),
        "dvn_r1cs_synthetic": (i32, [u64, u32, u32, C.POINTER(vp)]),
        "dvn_r1cs_info": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "dvn_r1cs_witness": (i32, [vp, vp]),
        "dvn_r1cs_free": (None, [vp]),
        "dvn_groth16_setup": (i32, [vp, vp, u64, C.POINTER(vp)]),
        "dvn_groth16_pk_info": (i32, [vp, C.POINTER(u32), C.POINTER(u64)]),
        "dvn_groth16_pk_free": (i32, [vp, vp]),
        "dvn_groth16_prove": (i32, [vp, vp, vp, u32, vp, vp, vp]),
        "dvn_groth16_last_ms": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "dvn_groth16_prove_shard": (i32, [vp, vp, vp, u32, u32, u32, vp]),
        "dvn_groth16_prove_combine": (i32, [vp, vp, u32, vp, vp, vp]),
        "dvn_groth16_proof_abi_encode": (i32, [vp, vp, vp, vp]),
        "dvn_pedersen_commit": (i32, [vp, vp, vp, vp, sz, u32, vp, vp]),
        "dvn_bls12377_points_upload_g1": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "dvn_bls12377_points_upload_g2": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "dvn_bls12377_points_synthetic_g1": (i32, [vp, u64, sz, C.POINTER(vp)]),
        "dvn_bls12377_points_download": (i32, [vp, vp, sz, sz, vp]),
        "dvn_bls12377_scalars_synthetic": (i32, [vp, u64, C.c_int, sz, vp]),
        "dvn_bls12377_msm_g1": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_bls12377_msm_g2": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_bls12377_ntt_domain": (i32, [vp, u32, C.POINTER(vp)]),
        "dvn_bls12377_ntt": (i32, [vp, vp, vp, C.c_int, C.c_int, C.c_int, u32]),
        "dvn_bls12377_groth16_compute_h": (i32, [vp, vp, vp, vp, vp, sz, vp, u32]),
        "dvn_bls12377_r1cs_synthetic": (i32, [u64, u32, u32, C.POINTER(vp)]),
        "dvn_bls12377_groth16_setup": (i32, [vp, vp, u64, C.POINTER(vp)]),
        "dvn_bls12377_groth16_prove": (i32, [vp, vp, vp, u32, vp, vp, vp]),
        "dvn_bls12381_points_upload_g1": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "dvn_bls12381_points_upload_g2": (i32, [vp, vp, sz, C.POINTER(vp)]),
        "dvn_bls12381_points_synthetic_g1": (i32, [vp, u64, sz, C.POINTER(vp)]),
        "dvn_bls12381_points_download": (i32, [vp, vp, sz, sz, vp]),
        "dvn_bls12381_scalars_synthetic": (i32, [vp, u64, C.c_int, sz, vp]),
        "dvn_bls12381_msm_g1": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_bls12381_msm_g2": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_bls12381_ntt_domain": (i32, [vp, u32, C.POINTER(vp)]),
        "dvn_bls12381_ntt": (i32, [vp, vp, vp, C.c_int, C.c_int, C.c_int, u32]),
        "dvn_bls12381_groth16_compute_h": (i32, [vp, vp, vp, vp, vp, sz, vp, u32]),
        "dvn_bls12381_r1cs_synthetic": (i32, [u64, u32, u32, C.POINTER(vp)]),
        "dvn_bls12381_groth16_setup": (i32, [vp, vp, u64, C.POINTER(vp)]),
        "dvn_bls12381_groth16_prove": (i32, [vp, vp, vp, u32, vp, vp, vp]),
        "dvn_groth16_vk_export": (i32, [vp, vp, sz, C.POINTER(sz)]),
        "dvn_groth16_pk_import": (i32, [vp, vp, C.POINTER(vp)]),
        "dvn_groth16_prove_solved": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]),
        "dvn_groth16_prove_solved_shard": (i32, [vp, vp, vp, vp, vp, vp, sz, u32, u32, vp]),
        "dvn_bls12377_groth16_prove_solved_shard": (i32, [vp, vp, vp, vp, vp, vp, sz, u32, u32, vp]),
        "dvn_bw6761_groth16_prove_solved_shard": (i32, [vp, vp, vp, vp, vp, vp, sz, u32, u32, vp]),
        "dvn_bls12377_groth16_prove_combine": (i32, [vp, vp, u32, vp, vp, vp]),
        "dvn_bw6761_groth16_prove_combine": (i32, [vp, vp, u32, vp, vp, vp]),
        "dvn_groth16_pk_export_arrays": (i32, [vp, vp, C.c_int, vp, sz, C.POINTER(sz)]),
        "dvn_bls12377_groth16_pk_import": (i32, [vp, vp, C.POINTER(vp)]),
        "dvn_bls12377_groth16_prove_solved": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]),
        "dvn_bls12377_groth16_pk_export_arrays": (i32, [vp, vp, C.c_int, vp, sz, C.POINTER(sz)]),
        "dvn_bls12381_groth16_pk_import": (i32, [vp, vp, C.POINTER(vp)]),
        "dvn_bls12381_groth16_prove_solved": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]),
        "dvn_bls12381_groth16_pk_export_arrays": (i32, [vp, vp, C.c_int, vp, sz, C.POINTER(sz)]),
        "dvn_bw6761_groth16_pk_import": (i32, [vp, vp, C.POINTER(vp)]),
        "dvn_bw6761_groth16_prove_solved": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp]),
        "dvn_bw6761_groth16_pk_export_arrays": (i32, [vp, vp, C.c_int, vp, sz, C.POINTER(sz)]),
        "dvn_r1cs_synthetic_committed": (i32, [u64, u32, u32, u32, C.POINTER(vp)]),
        "dvn_groth16_prove_committed": (i32, [vp, vp, vp, u32, vp, vp, vp, vp, vp, vp]),
        "dvn_groth16_prove_committed_shard": (i32, [vp, vp, vp, u32, u32, u32, vp, vp, vp, vp]),
        "dvn_groth16_prove_committed_multi": (i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]),
        "dvn_groth16_proof_words": (i32, [vp, C.POINTER(sz)]),
        "dvn_keccak256": (i32, [vp, sz, vp]),
        "dvn_kzg_setup_load": (i32, [vp, vp, vp, C.POINTER(vp)]),
        "dvn_kzg_setup_free": (i32, [vp, vp]),
        "dvn_kzg_blob_to_commitment": (i32, [vp, vp, vp, vp]),
        "dvn_kzg_compute_proof": (i32, [vp, vp, vp, vp, vp, vp]),
        "dvn_kzg_compute_cell_proofs": (i32, [vp, vp, vp, vp]),
        "dvn_kzg_compute_blob_proof": (i32, [vp, vp, vp, vp, vp]),
        "dvn_sha256": (i32, [vp, sz, vp]),
        "dvn_kzg_evaluate_barycentric": (i32, [vp, vp, vp, vp, vp]),
        "dvn_r1cs_curve": (i32, [vp]),
        "dvn_bw6761_points_upload": (i32, [vp, C.c_int, vp, sz, C.POINTER(vp)]),
        "dvn_bw6761_points_synthetic": (i32, [vp, C.c_int, vp, u64, sz, C.POINTER(vp)]),
        "dvn_bw6761_points_download": (i32, [vp, vp, sz, sz, vp]),
        "dvn_bw6761_scalars_synthetic": (i32, [vp, u64, C.c_int, sz, vp]),
        "dvn_bw6761_msm": (i32, [vp, vp, sz, sz, vp, u32, vp]),
        "dvn_bw6761_ntt_domain": (i32, [vp, u32, C.POINTER(vp)]),
        "dvn_bw6761_ntt": (i32, [vp, vp, vp, C.c_int, C.c_int, C.c_int, u32]),
        "dvn_bw6761_groth16_compute_h": (i32, [vp, vp, vp, vp, vp, sz, vp, u32]),
        "dvn_bw6761_r1cs_synthetic": (i32, [u64, u32, u32, C.POINTER(vp)]),
        "dvn_bw6761_groth16_setup": (i32, [vp, vp, u64, C.POINTER(vp)]),
        "dvn_bw6761_groth16_prove": (i32, [vp, vp, vp, u32, vp, vp, vp]),
        "dvn_ctx_device_count": (i32, [vp]),
        "dvn_ctx_device": (vp, [vp, i32]),
        "dvn_ctx_device_ordinal": (i32, [vp]),
        "dvn_error_string": (C.c_char_p, [i32]),
        "dvn_msm_g1_share_submit": (i32, [vp, vp, sz, sz, vp, u32, u32, u32, C.POINTER(i32)]),
        "dvn_msm_g1_share_collect": (i32, [vp, i32, vp]),
        "dvn_msm_g1_sum": (i32, [vp, u32, vp]),
        "dvn_msm_g1_multi": (i32, [vp, vp, u32, sz, sz, vp, vp]),
        "dvn_comm_unique_id": (i32, [vp]),
        "dvn_comm_init": (i32, [vp, vp, u32, u32, C.POINTER(vp)]),
        "dvn_comm_all_gather": (i32, [vp, vp, vp, sz, vp]),
        "dvn_comm_free": (i32, [vp, vp]),
        "dvn_groth16_prove_multi": (i32, [vp, vp, u32, vp, vp, vp, vp]),
        "dvn_groth16_prove_solved_multi": (i32, [vp, vp, u32, vp, vp, vp, vp, sz, vp, vp, vp]),
        "dvn_bw6761_groth16_prove_solved_multi": (i32, [vp, vp, u32, vp, vp, vp, vp, sz, vp, vp, vp]),
        "dvn_groth16_last_stats_json": (i32, [vp, vp, sz, C.POINTER(sz)]),
        "dvn_bls12377_pedersen_commit": (i32, [vp, vp, vp, vp, sz, u32, vp, vp]),
        "dvn_bls12381_pedersen_commit": (i32, [vp, vp, vp, vp, sz, u32, vp, vp]),
        "dvn_bw6761_pedersen_commit": (i32, [vp, vp, vp, vp, sz, u32, vp, vp]),
        "dvn_gnark_pk_scan": (i32, [vp, sz, vp]),
        "dvn_groth16_pk_load_bytes": (i32, [vp, vp, sz, u32, vp, u32, C.POINTER(vp)]),
        "dvn_groth16_pk_load_file": (i32, [vp, C.c_char_p, C.c_char_p, u32, vp, u32, C.POINTER(vp)]),
        "dvn_groth16_pk_evict": (i32, [vp, vp]),
        "dvn_bls12377_gnark_pk_scan": (i32, [vp, sz, vp]),
        "dvn_bls12377_groth16_pk_load_bytes": (i32, [vp, vp, sz, u32, vp, u32, C.POINTER(vp)]),
        "dvn_bls12377_groth16_pk_load_file": (i32, [vp, C.c_char_p, C.c_char_p, u32, vp, u32, C.POINTER(vp)]),
        "dvn_bw6761_gnark_pk_scan": (i32, [vp, sz, vp]),
        "dvn_bw6761_groth16_pk_load_bytes": (i32, [vp, vp, sz, u32, vp, u32, C.POINTER(vp)]),
        "dvn_bw6761_groth16_pk_load_file": (i32, [vp, C.c_char_p, C.c_char_p, u32, vp, u32, C.POINTER(vp)]),
        "dvn_groth16_pk_commitment_bases": (i32, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "dvn_prover_use_gpu": (i32, []),
        "dvn_prover_set_use_gpu": (None, [i32]),
        "dvn_prover_reload_env": (None, []),
        "dvn_prover_set_prover": (None, [vp, vp]),
        "dvn_prover_set_cpu_backend": (None, [vp, vp]),
        "dvn_prover_last_backend": (i32, []),
        "dvn_prover_prove": (i32, [vp, i32, vp, vp, vp, vp, vp, vp]),
        "dvn_prover_gpu_prover": (i32, [vp, i32, vp, vp, vp, vp, vp, vp]),
        "dvn_prover_cpu_prover": (i32, [vp, i32, vp, vp, vp, vp, vp, vp]),
        "dvn_prover_setup": (i32, [vp, vp, u64, C.POINTER(vp)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)           # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "dvn_init", "dvn_shutdown", "dvn_last_error", "dvn_version", "dvn_test_raise", "dvn_bench_mad_rate", "dvn_malloc", "dvn_free", "dvn_memcpy_h2d",
    "dvn_memcpy_d2h", "dvn_synchronize", "dvn_points_upload_g1", "dvn_points_upload_g2", "dvn_points_synthetic_g1",
    "dvn_points_synthetic_g2", "dvn_scalars_synthetic", "dvn_points_set_expanded", "dvn_points_download", "dvn_points_len", "dvn_points_free", "dvn_msm_g1", "dvn_msm_g1_static", "dvn_msm_g2", "dvn_msm_g1_batched",
    "dvn_msm_g1_submit", "dvn_msm_g1_collect", "dvn_msm_g2_submit", "dvn_msm_g2_collect", "dvn_msm_plan", "dvn_msm_plan_points", "dvn_msm_g1_windows", "dvn_msm_g1_windows_submit", "dvn_msm_g1_windows_collect", "dvn_msm_combine_g1", "dvn_ntt_domain", "dvn_ntt_domain_free", "dvn_ntt",
    "dvn_groth16_compute_h", "dvn_last_kernel_ms", "dvn_r1cs_synthetic", "dvn_r1cs_info", "dvn_r1cs_witness", "dvn_r1cs_free",
    "dvn_groth16_setup", "dvn_groth16_pk_info", "dvn_groth16_pk_free", "dvn_groth16_prove", "dvn_groth16_last_ms", "dvn_groth16_prove_shard", "dvn_groth16_prove_combine",
    "dvn_groth16_proof_abi_encode", "dvn_pedersen_commit",
    "dvn_bls12377_points_upload_g1", "dvn_bls12377_points_upload_g2", "dvn_bls12377_points_synthetic_g1", "dvn_bls12377_points_download",
    "dvn_bls12377_scalars_synthetic", "dvn_bls12377_msm_g1", "dvn_bls12377_msm_g2", "dvn_bls12377_ntt_domain", "dvn_bls12377_ntt", "dvn_bls12377_groth16_compute_h",
    "dvn_bls12377_r1cs_synthetic", "dvn_bls12377_groth16_setup", "dvn_bls12377_groth16_prove",
    "dvn_bls12381_points_upload_g1", "dvn_bls12381_points_upload_g2", "dvn_bls12381_points_synthetic_g1", "dvn_bls12381_points_download",
    "dvn_bls12381_scalars_synthetic", "dvn_bls12381_msm_g1", "dvn_bls12381_msm_g2", "dvn_bls12381_ntt_domain", "dvn_bls12381_ntt", "dvn_bls12381_groth16_compute_h",
    "dvn_bls12381_r1cs_synthetic", "dvn_bls12381_groth16_setup", "dvn_bls12381_groth16_prove", "dvn_r1cs_curve",
    "dvn_r1cs_synthetic_committed", "dvn_groth16_prove_committed", "dvn_groth16_prove_committed_shard", "dvn_groth16_prove_committed_multi", "dvn_groth16_proof_words", "dvn_keccak256", "dvn_groth16_vk_export", "dvn_groth16_pk_import", "dvn_groth16_prove_solved", "dvn_groth16_prove_solved_shard", "dvn_bls12377_groth16_prove_solved_shard", "dvn_bw6761_groth16_prove_solved_shard",
    "dvn_bls12377_groth16_prove_combine", "dvn_bw6761_groth16_prove_combine", "dvn_groth16_pk_export_arrays",
    "dvn_bls12377_groth16_pk_import", "dvn_bls12377_groth16_prove_solved", "dvn_bls12377_groth16_pk_export_arrays", "dvn_bls12381_groth16_pk_import", "dvn_bls12381_groth16_prove_solved", "dvn_bls12381_groth16_pk_export_arrays", "dvn_bw6761_groth16_pk_import", "dvn_bw6761_groth16_prove_solved", "dvn_bw6761_groth16_pk_export_arrays",
    "dvn_kzg_setup_load", "dvn_kzg_setup_free", "dvn_kzg_blob_to_commitment", "dvn_kzg_compute_proof", "dvn_kzg_compute_cell_proofs",
    "dvn_kzg_evaluate_barycentric", "dvn_kzg_compute_blob_proof", "dvn_sha256",
    "dvn_bw6761_points_upload", "dvn_bw6761_points_synthetic", "dvn_bw6761_points_download", "dvn_bw6761_scalars_synthetic", "dvn_bw6761_msm",
    "dvn_bw6761_ntt_domain", "dvn_bw6761_ntt", "dvn_bw6761_groth16_compute_h",
    "dvn_bw6761_r1cs_synthetic", "dvn_bw6761_groth16_setup", "dvn_bw6761_groth16_prove",
    "dvn_prover_use_gpu", "dvn_prover_set_use_gpu", "dvn_prover_reload_env", "dvn_prover_set_prover", "dvn_prover_set_cpu_backend",
    "dvn_prover_last_backend", "dvn_prover_prove", "dvn_prover_gpu_prover", "dvn_prover_cpu_prover", "dvn_prover_setup",
    "dvn_ctx_device_count", "dvn_ctx_device", "dvn_ctx_device_ordinal", "dvn_error_string",
    "dvn_msm_g1_share_submit", "dvn_msm_g1_share_collect", "dvn_msm_g1_sum", "dvn_msm_g1_multi",
    "dvn_comm_unique_id", "dvn_comm_init", "dvn_comm_all_gather", "dvn_comm_free",
    "dvn_groth16_prove_multi", "dvn_groth16_prove_solved_multi", "dvn_bw6761_groth16_prove_solved_multi", "dvn_groth16_last_stats_json",
    "dvn_bls12377_pedersen_commit", "dvn_bls12381_pedersen_commit", "dvn_bw6761_pedersen_commit",
    "dvn_gnark_pk_scan", "dvn_groth16_pk_load_bytes", "dvn_groth16_pk_load_file", "dvn_groth16_pk_commitment_bases", "dvn_groth16_pk_evict",
    "dvn_bls12377_gnark_pk_scan", "dvn_bls12377_groth16_pk_load_bytes", "dvn_bls12377_groth16_pk_load_file",
    "dvn_bw6761_gnark_pk_scan", "dvn_bw6761_groth16_pk_load_bytes", "dvn_bw6761_groth16_pk_load_file",
]

# (curve, ccs, pk, witness, r, s, proof, user): what the reference's CPUProverWithWitness receives
CPU_BACKEND_T = C.CFUNCTYPE(C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
PROVER_FUNC_T = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p)
BN254, BLS12_377, BLS12_381, BW6_761 = 1, 2, 3, 4


def _host(a: np.ndarray) -> C.c_void_p:
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "expect C-contiguous uint64"
    return C.c_void_p(a.ctypes.data)


class Points:
    def __init__(self, ctx: "Context", handle, curve: int):
        self.ctx, self.handle, self.curve = ctx, handle, curve

    def __len__(self):
        return int(self.ctx.L.dvn_points_len(self.handle))

    def download(self, off: int = 0, n: int | None = None) -> np.ndarray:
        n = len(self) - off if n is None else n
        out = np.zeros((n, 8 if self.curve == 1 else 16), dtype=np.uint64)
        self.ctx._chk(self.ctx.L.dvn_points_download(self.ctx.h, self.handle, off, n, _host(out)))
        return out

    def set_expanded(self, window_bits: int):
        """Rebuild as an expanded table (row w = 2^(window_bits w) P); 0 collapses it."""
        self.ctx._chk(self.ctx.L.dvn_points_set_expanded(self.ctx.h, self.handle, window_bits))

    def free(self):
        if self.handle:
            self.ctx._chk(self.ctx.L.dvn_points_free(self.ctx.h, self.handle))
            self.handle = None


class Domain:
    def __init__(self, ctx: "Context", handle, log_n: int):
        self.ctx, self.handle, self.log_n = ctx, handle, log_n

    @property
    def n(self):
        return 1 << self.log_n

    def free(self):
        if self.handle:
            self.ctx._chk(self.ctx.L.dvn_ntt_domain_free(self.ctx.h, self.handle))
            self.handle = None


class Context:
    """One context per process-GPU (dvn_init).  Raises DvnError on any failure; never falls back to a CPU path."""

    def __init__(self, device=0, _handle=None, _owner=None):
        """device: one HIP ordinal, or a list of ordinals (dvn_init(devs, ndev)): the context then serves devs[0] and
        .device(i) is the context of devs[i]; an ordinal may repeat (two contexts on one GPU)."""
        self.L = load_library()
        self._owner = _owner
        if _handle is not None:
            self.h = _handle
            return
        devices = list(device) if isinstance(device, (list, tuple)) else [device]
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self.L.dvn_init(devs, len(devices), C.byref(h))
        if rc != 0:
            raise DvnError(rc, self.L.dvn_last_error(None).decode())
        self.h = h

    def close(self):
        if self.h and self._owner is None:
            self.L.dvn_shutdown(self.h)
        self.h = None

    def device_count(self) -> int:
        return int(self.L.dvn_ctx_device_count(self.h))

    def device(self, index: int) -> "Context":
        """The context of the index-th device of a multi-device context (0 = this one); owned by this context."""
        if index == 0:
            return self
        h = self.L.dvn_ctx_device(self.h, index)
        if not h:
            raise DvnError(-1, self.L.dvn_last_error(self.h).decode())
        return Context(_handle=C.c_void_p(h), _owner=self)

    def _chk(self, rc: int):
        if rc != 0:
            raise DvnError(rc, self.L.dvn_last_error(self.h).decode())

    # ---- memory
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p(); self._chk(self.L.dvn_malloc(self.h, nbytes, C.byref(p))); return p.value

    def free(self, dptr: int):
        self._chk(self.L.dvn_free(self.h, C.c_void_p(dptr)))

    def h2d(self, dptr: int, a: np.ndarray):
        self._chk(self.L.dvn_memcpy_h2d(self.h, C.c_void_p(dptr), C.c_void_p(a.ctypes.data), a.nbytes))

    def d2h(self, a: np.ndarray, dptr: int):
        self._chk(self.L.dvn_memcpy_d2h(self.h, C.c_void_p(a.ctypes.data), C.c_void_p(dptr), a.nbytes))

    def to_device(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a); p = self.malloc(a.nbytes); self.h2d(p, a); return p

    def synchronize(self):
        self._chk(self.L.dvn_synchronize(self.h))

    # ---- points
    def upload_g1(self, xy: np.ndarray) -> Points:
        xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 8)
        p = C.c_void_p(); self._chk(self.L.dvn_points_upload_g1(self.h, _host(xy), xy.shape[0], C.byref(p))); return Points(self, p, 1)

    def upload_g2(self, xy: np.ndarray) -> Points:
        xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 16)
        p = C.c_void_p(); self._chk(self.L.dvn_points_upload_g2(self.h, _host(xy), xy.shape[0], C.byref(p))); return Points(self, p, 2)

    def synthetic_g1(self, seed: int, n: int) -> Points:
        p = C.c_void_p(); self._chk(self.L.dvn_points_synthetic_g1(self.h, seed, n, C.byref(p))); return Points(self, p, 1)

    def synthetic_g2(self, seed: int, n: int) -> Points:
        p = C.c_void_p(); self._chk(self.L.dvn_points_synthetic_g2(self.h, seed, n, C.byref(p))); return Points(self, p, 2)

    def synthetic_scalars(self, seed: int, kind: int, n: int) -> int:
        """Device-resident synthetic scalar vector (n x 32 B); returns the device pointer (free with ctx.free)."""
        d = self.malloc(n * 32)
        self._chk(self.L.dvn_scalars_synthetic(self.h, seed, kind, n, C.c_void_p(d)))
        return d

    # ---- MSM.  scalars: (n,4) uint64 host array, or an int device pointer with n given.
    def _scalars(self, scalars, n):
        if isinstance(scalars, (int, np.integer)):
            assert n is not None
            return C.c_void_p(int(scalars)), n, MSM_SCALARS_ON_DEVICE
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        self._keep = scalars
        return _host(scalars), scalars.shape[0], 0

    def msm_g1(self, pts: Points, scalars, off: int = 0, n: int | None = None) -> np.ndarray:
        """Returns the G1Jac image (12 uint64).  n defaults to len(scalars); MultiExp's length rule is checked by the library."""
        sp, ns, fl = self._scalars(scalars, n)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1(self.h, pts.handle, off, ns, sp, fl, _host(out)))
        return out

    def msm_g1_static(self, pts: Points, d_scalars: int, n: int, off: int = 0, flags: int = MSM_SCALARS_ON_DEVICE) -> np.ndarray:
        """dvn_msm_g1 over the library's static scratch buffer (no allocation): device scalars, n <= MSM_STATIC_MAX_N."""
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1_static(self.h, pts.handle, off, n, C.c_void_p(int(d_scalars)), flags, _host(out)))
        return out

    def msm_g1_batched(self, pts: Points, scalars: np.ndarray, off: int = 0) -> np.ndarray:
        """scalars: (batch, n, 4) uint64 host array -> (batch, 12) G1Jac images, one pass over the shared point set."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        assert scalars.ndim == 3 and scalars.shape[2] == 4
        batch, n = scalars.shape[0], scalars.shape[1]
        out = np.zeros((batch, 12), dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1_batched(self.h, pts.handle, off, n, _host(scalars), batch, 0, _host(out)))
        return out

    def msm_g1_submit(self, pts: Points, scalars, off: int = 0, n: int | None = None) -> int:
        """Queue a G1 MSM on a free lane; returns the ticket for msm_g1_collect (at most four outstanding: one per lane)."""
        sp, ns, fl = self._scalars(scalars, n)
        t = C.c_int32()
        self._chk(self.L.dvn_msm_g1_submit(self.h, pts.handle, off, ns, sp, fl, C.byref(t)))
        return t.value

    def msm_g1_collect(self, ticket: int) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1_collect(self.h, ticket, _host(out)))
        return out

    def msm_g2_submit(self, pts: Points, scalars, off: int = 0, n: int | None = None) -> int:
        sp, ns, fl = self._scalars(scalars, n)
        t = C.c_int32()
        self._chk(self.L.dvn_msm_g2_submit(self.h, pts.handle, off, ns, sp, fl, C.byref(t)))
        return t.value

    def msm_g2_collect(self, ticket: int) -> np.ndarray:
        out = np.zeros(24, dtype=np.uint64)
        self._chk(self.L.dvn_msm_g2_collect(self.h, ticket, _host(out)))
        return out

    def msm_g2(self, pts: Points, scalars, off: int = 0, n: int | None = None) -> np.ndarray:
        sp, ns, fl = self._scalars(scalars, n)
        out = np.zeros(24, dtype=np.uint64)
        self._chk(self.L.dvn_msm_g2(self.h, pts.handle, off, ns, sp, fl, _host(out)))
        return out

    def msm_plan(self, n: int):
        c, w = C.c_uint32(), C.c_uint32()
        rc = self.L.dvn_msm_plan(n, C.byref(c), C.byref(w))
        self._chk(rc)
        return c.value, w.value

    def msm_plan_points(self, pts: Points, n: int):
        """(window_bits, nwin, combine_bits) for an n-scalar MSM over this point set."""
        c, w, cb = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.dvn_msm_plan_points(pts.handle, n, C.byref(c), C.byref(w), C.byref(cb)))
        return c.value, w.value, cb.value

    def msm_g1_windows(self, pts: Points, scalars, win_lo: int, win_hi: int, off: int = 0, n: int | None = None) -> np.ndarray:
        sp, ns, fl = self._scalars(scalars, n)
        out = np.zeros((win_hi - win_lo, 16), dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1_windows(self.h, pts.handle, off, ns, sp, fl, win_lo, win_hi, _host(out)))
        return out

    def msm_g1_windows_submit(self, pts: Points, scalars, win_lo: int, win_hi: int, off: int = 0, n: int | None = None):
        sp, ns, fl = self._scalars(scalars, n)
        t = C.c_int32()
        self._chk(self.L.dvn_msm_g1_windows_submit(self.h, pts.handle, off, ns, sp, fl, win_lo, win_hi, C.byref(t)))
        return t.value, win_hi - win_lo

    def msm_g1_windows_collect(self, ticket) -> np.ndarray:
        t, w = ticket
        out = np.zeros((w, 16), dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1_windows_collect(self.h, t, _host(out)))
        return out

    def msm_g1_share_submit(self, pts: Points, d_scalars: int, n: int, rank: int, world: int, off: int = 0) -> int:
        """Queue this rank's share of an MSM shared by entry count (device scalars); returns the ticket."""
        t = C.c_int32()
        self._chk(self.L.dvn_msm_g1_share_submit(self.h, pts.handle, off, n, C.c_void_p(int(d_scalars)), MSM_SCALARS_ON_DEVICE, rank, world, C.byref(t)))
        return t.value

    def msm_g1_share_collect(self, ticket: int) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1_share_collect(self.h, ticket, _host(out)))
        return out

    def msm_g1_sum(self, jacs: np.ndarray) -> np.ndarray:
        """Host only: the sum of G1Jac images (count, 12) -- the shares of all ranks add up to the MSM."""
        jacs = np.ascontiguousarray(jacs, dtype=np.uint64).reshape(-1, 12)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.dvn_msm_g1_sum(_host(jacs), jacs.shape[0], _host(out)))
        return out

    def msm_combine_g1(self, window_sums: np.ndarray, window_bits: int) -> np.ndarray:
        ws = np.ascontiguousarray(window_sums, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.dvn_msm_combine_g1(self.h, _host(ws), ws.shape[0], window_bits, _host(out)))
        return out

    # ---- NTT
    def domain(self, log_n: int) -> Domain:
        p = C.c_void_p(); self._chk(self.L.dvn_ntt_domain(self.h, log_n, C.byref(p))); return Domain(self, p, log_n)

    def ntt(self, dom: Domain, data, inverse: bool, decimation: int, coset: bool = False):
        """In place.  data: (n,4) uint64 host array (modified) or int device pointer."""
        if isinstance(data, (int, np.integer)):
            self._chk(self.L.dvn_ntt(self.h, dom.handle, C.c_void_p(int(data)), int(inverse), decimation, int(coset), NTT_DATA_ON_DEVICE))
            return data
        assert data.dtype == np.uint64 and data.flags["C_CONTIGUOUS"] and data.shape == (dom.n, 4)
        self._chk(self.L.dvn_ntt(self.h, dom.handle, _host(data), int(inverse), decimation, int(coset), 0))
        return data

    def compute_h(self, dom: Domain, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
        a, b, c = (np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4) for v in (a, b, c))
        if not (a.shape == b.shape == c.shape):
            raise ValueError("a, b, c must have the same length")
        out = np.zeros((dom.n, 4), dtype=np.uint64)
        self._chk(self.L.dvn_groth16_compute_h(self.h, dom.handle, _host(a), _host(b), _host(c), a.shape[0], _host(out), 0))
        return out

    def compute_h_device(self, dom: Domain, da: int, db: int, dc: int, length: int, dout: int):
        """computeH on device-resident a, b, c (`length` elements each, left untouched) into the device buffer dout (dom.n
        elements); last_kernel_ms(0) is the device time of the seven transforms and the pointwise pass."""
        vp = C.c_void_p
        self._chk(self.L.dvn_groth16_compute_h(self.h, dom.handle, vp(da), vp(db), vp(dc), length, vp(dout), 1))

    def pedersen_commit(self, basis: Points, basis_exp_sigma: Points, values: np.ndarray):
        """(commitment, pok) G1Affine images: MSM(Basis, values), MSM(BasisExpSigma, values)."""
        values = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        c, k = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        self._chk(self.L.dvn_pedersen_commit(self.h, basis.handle, basis_exp_sigma.handle, _host(values), values.shape[0], 0, _host(c), _host(k)))
        return c, k

    def mad_rate(self) -> float:
        """Chip-wide v_mad_u64_u32 lane operations per second of this device, measured now (dvn_bench_mad_rate)."""
        v = C.c_double()
        self._chk(self.L.dvn_bench_mad_rate(self.h, C.byref(v)))
        return float(v.value)

    def last_kernel_ms(self, which: int = 0) -> float:
        ms = C.c_float(); self._chk(self.L.dvn_last_kernel_ms(self.h, which, C.byref(ms))); return float(ms.value)


class R1CS:
    """Host-side sparse R1CS handle (dvn_r1cs): the synthetic multiplication-gate instance."""

    def __init__(self, seed: int, n_constraints: int, n_public: int, curve: str = "bn254", n_committed: int = 0):
        self.L = load_library()
        self.curve = curve
        self.n_committed = n_committed
        h = C.c_void_p()
        if n_committed:
            assert curve == "bn254", "committed instances exist on BN254 (the statetransition curve)"
            rc = self.L.dvn_r1cs_synthetic_committed(seed, n_constraints, n_public, n_committed, C.byref(h))
        else:
            fn = self.L.dvn_r1cs_synthetic if curve == "bn254" else getattr(self.L, f"dvn_{curve}_r1cs_synthetic")
            rc = fn(seed, n_constraints, n_public, C.byref(h))
        if rc != 0:
            raise DvnError(rc, self.L.dvn_last_error(None).decode())
        self.h = h
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.dvn_r1cs_info(h, C.byref(a), C.byref(b), C.byref(c))
        self.n_wires, self.n_public, self.n_constraints = a.value, b.value, c.value

    def witness(self) -> np.ndarray:
        out = np.zeros((self.n_wires, 6 if self.curve == "bw6761" else 4), dtype=np.uint64)
        self.L.dvn_r1cs_witness(self.h, _host(out))
        return out

    def free(self):
        if self.h:
            self.L.dvn_r1cs_free(self.h)
            self.h = None


class Groth16Key:
    """Device-resident proving key (dvn_groth16_pk) produced by the Setup mirror."""

    def __init__(self, ctx: Context, cs: R1CS, seed: int):
        self.ctx = ctx
        self.curve = cs.curve
        self._prefix = "dvn_" if cs.curve == "bn254" else f"dvn_{cs.curve}_"
        self._a1 = {"bn254": 8, "bw6761": 24}.get(cs.curve, 12)      # uint64 words of a G1Affine image
        self._a2 = self._a1 if cs.curve == "bw6761" else 2 * self._a1  # G2Affine: twice that, except BW6-761 (G2 over Fp)
        h = C.c_void_p()
        ctx._chk(getattr(ctx.L, self._prefix + "groth16_setup")(ctx.h, cs.h, seed, C.byref(h)))
        self.h = h
        ln = C.c_uint32(); sizes = (C.c_uint64 * 5)()
        ctx.L.dvn_groth16_pk_info(h, C.byref(ln), sizes)
        self.log_n = ln.value
        self.sizes = dict(zip(("A", "B", "K", "Z", "B2"), (int(v) for v in sizes)))

    def prove(self, witness, r_mont: np.ndarray, s_mont: np.ndarray) -> dict:
        """witness: (m,4) uint64 host array or int device pointer.  Returns {'Ar': (8,), 'Bs': (16,), 'Krs': (8,)}."""
        a1, a2 = self._a1, self._a2
        out = np.zeros(2 * a1 + a2, dtype=np.uint64)
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
        if isinstance(witness, (int, np.integer)):
            wp, fl = C.c_void_p(int(witness)), 1
        else:
            witness = np.ascontiguousarray(witness, dtype=np.uint64); wp, fl = _host(witness), 0
        self.ctx._chk(getattr(self.ctx.L, self._prefix + "groth16_prove")(self.ctx.h, self.h, wp, fl, _host(r_mont), _host(s_mont), _host(out)))
        return {"Ar": out[0:a1].copy(), "Bs": out[a1:a1 + a2].copy(), "Krs": out[a1 + a2:2 * a1 + a2].copy()}

    def verifying_key(self, n_public: int, committed: bool = False) -> dict:
        """The vk half of Setup: {'alpha': (a1,), 'beta'/'gamma'/'delta': (a2,), 'ic': (n_public, a1)[, 'pedersen_g', 'pedersen_gsigmaneg']}."""
        n = C.c_size_t()
        self.ctx._chk(self.ctx.L.dvn_groth16_vk_export(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        self.ctx._chk(self.ctx.L.dvn_groth16_vk_export(self.h, _host(out), n.value, C.byref(n)))
        a1, a2 = self._a1, self._a2
        vk = {"alpha": out[0:a1], "beta": out[a1:a1 + a2], "gamma": out[a1 + a2:a1 + 2 * a2], "delta": out[a1 + 2 * a2:a1 + 3 * a2]}
        at = a1 + 3 * a2
        vk["ic"] = out[at:at + n_public * a1].reshape(n_public, a1); at += n_public * a1
        if committed:
            vk["pedersen_g"] = out[at:at + a2]; vk["pedersen_gsigmaneg"] = out[at + a2:at + 2 * a2]
        return vk

    def export_arrays(self, n_wires: int, n_public: int) -> dict:
        """The key's arrays in the layout dvn_<curve>_groth16_pk_import takes (keys without commitment)."""
        fn = getattr(self.ctx.L, self._prefix + "groth16_pk_export_arrays")
        a1, a2 = self._a1, self._a2
        def get(which, dtype=np.uint64):
            n = C.c_size_t()
            self.ctx._chk(fn(self.ctx.h, self.h, which, None, 0, C.byref(n)))
            out = np.zeros(max(n.value, 1), dtype=np.uint64)
            self.ctx._chk(fn(self.ctx.h, self.h, which, _host(out), out.size, C.byref(n)))
            return out[:n.value]
        arr = {"A": get(0).reshape(-1, a1), "B": get(1).reshape(-1, a1), "K": get(2).reshape(-1, a1), "Z": get(3).reshape(-1, a1),
               "B2": get(4).reshape(-1, a2), "consts": get(7)}
        arr["idxA"] = get(5).view(np.uint32)[:self.sizes["A"]].copy()
        arr["idxB"] = get(6).view(np.uint32)[:self.sizes["B"]].copy()
        arr["idxK"] = np.arange(n_public, n_wires, dtype=np.uint32)
        return arr

    def prove_committed(self, witness, r_mont: np.ndarray, s_mont: np.ndarray) -> dict:
        """Proof with one BSB22 commitment (BN254): adds 'Commitment', 'CommitmentPok' (8 uint64 each) and 'challenge' (4)."""
        assert self.curve == "bn254"
        out = np.zeros(32, dtype=np.uint64); cm = np.zeros(8, dtype=np.uint64); pok = np.zeros(8, dtype=np.uint64); ch = np.zeros(4, dtype=np.uint64)
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
        if isinstance(witness, (int, np.integer)):
            wp, fl = C.c_void_p(int(witness)), 1
        else:
            witness = np.ascontiguousarray(witness, dtype=np.uint64); wp, fl = _host(witness), 0
        self.ctx._chk(self.ctx.L.dvn_groth16_prove_committed(self.ctx.h, self.h, wp, fl, _host(r_mont), _host(s_mont), _host(out), _host(cm), _host(pok), _host(ch)))
        return {"Ar": out[0:8].copy(), "Bs": out[8:24].copy(), "Krs": out[24:32].copy(), "Commitment": cm, "CommitmentPok": pok, "challenge": ch}

    def prove_committed_shard(self, witness, rank: int, world: int) -> dict:
        """Rank `rank` of `world` of a committed proof: the commitment hint / challenge / witness completion whole, then this
        rank's share of the five MSMs.  'partial': (72,) uint64 to be all-gathered; the other entries are the same on all ranks."""
        assert self.curve == "bn254"
        out = np.zeros(72, dtype=np.uint64); cm = np.zeros(8, dtype=np.uint64); pok = np.zeros(8, dtype=np.uint64); ch = np.zeros(4, dtype=np.uint64)
        if isinstance(witness, (int, np.integer)):
            wp, fl = C.c_void_p(int(witness)), 1
        else:
            witness = np.ascontiguousarray(witness, dtype=np.uint64); wp, fl = _host(witness), 0
        self.ctx._chk(self.ctx.L.dvn_groth16_prove_committed_shard(self.ctx.h, self.h, wp, fl, rank, world, _host(out), _host(cm), _host(pok), _host(ch)))
        return {"partial": out, "Commitment": cm, "CommitmentPok": pok, "challenge": ch}

    def proof_words(self) -> int:
        n = C.c_size_t()
        self.ctx._chk(self.ctx.L.dvn_groth16_proof_words(self.h, C.byref(n)))
        return int(n.value)

    def prove_shard(self, witness, rank: int, world: int) -> np.ndarray:
        """This rank's partial results of the five MSMs (BN254 keys): (72,) uint64 to be all-gathered."""
        assert self.curve == "bn254"
        out = np.zeros(72, dtype=np.uint64)
        if isinstance(witness, (int, np.integer)):
            wp, fl = C.c_void_p(int(witness)), 1
        else:
            witness = np.ascontiguousarray(witness, dtype=np.uint64); wp, fl = _host(witness), 0
        self.ctx._chk(self.ctx.L.dvn_groth16_prove_shard(self.ctx.h, self.h, wp, fl, rank, world, _host(out)))
        return out

    def prove_combine(self, partials: np.ndarray, r_mont: np.ndarray, s_mont: np.ndarray) -> dict:
        """partials: (world, 72) uint64 as gathered from the ranks.  Host only."""
        partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, 72)
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
        out = np.zeros(32, dtype=np.uint64)
        self.ctx._chk(self.ctx.L.dvn_groth16_prove_combine(self.h, _host(partials), partials.shape[0], _host(r_mont), _host(s_mont), _host(out)))
        return {"Ar": out[0:8].copy(), "Bs": out[8:24].copy(), "Krs": out[24:32].copy()}

    def last_ms(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        self.ctx.L.dvn_groth16_last_ms(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"compute_h": a.value, "msm": b.value, "total": c.value}

    def last_stats(self) -> dict:
        """dvn_groth16_last_stats_json of the last prove call on this key, parsed."""
        return _stats_json(self.ctx, self.h)

    def free(self):
        if self.h:
            self.ctx._chk(self.ctx.L.dvn_groth16_pk_free(self.ctx.h, self.h))
            self.h = None


def _stats_json(ctx, key_handle) -> dict:
    import json
    need = C.c_size_t()
    ctx._chk(ctx.L.dvn_groth16_last_stats_json(key_handle, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    ctx._chk(ctx.L.dvn_groth16_last_stats_json(key_handle, buf, need.value, C.byref(need)))
    return json.loads(buf.value.decode())


def prove_multi(ctx: "Context", keys, witness: np.ndarray, r_mont, s_mont, solved=None, curve: str = "bn254") -> dict:
    """One proof over all devices of a multi-device context.  keys: one key handle object (Groth16Key / ImportedKey /
    LoadedKey) per device, keys[i] resident on ctx.device(i).  solved = (a, b, c): prove from the solver's output."""
    a1 = {"bn254": 8, "bw6761": 24}.get(curve, 12)
    a2 = a1 if curve == "bw6761" else 2 * a1
    arr = (C.c_void_p * len(keys))(*[k.h for k in keys])
    witness = np.ascontiguousarray(witness, dtype=np.uint64)
    r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
    out = np.zeros(2 * a1 + a2, dtype=np.uint64)
    if solved is None:
        assert curve == "bn254"
        ctx._chk(ctx.L.dvn_groth16_prove_multi(ctx.h, arr, len(keys), _host(witness), _host(r_mont), _host(s_mont), _host(out)))
    else:
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint64) for x in solved)
        fn = ctx.L.dvn_groth16_prove_solved_multi if curve == "bn254" else getattr(ctx.L, f"dvn_{curve}_groth16_prove_solved_multi")
        ctx._chk(fn(ctx.h, arr, len(keys), _host(witness), _host(a), _host(b), _host(c), a.shape[0], _host(r_mont), _host(s_mont), _host(out)))
    return {"Ar": out[0:a1].copy(), "Bs": out[a1:a1 + a2].copy(), "Krs": out[a1 + a2:2 * a1 + a2].copy()}


def msm_g1_multi(ctx: "Context", sets, scalars: np.ndarray, off: int = 0) -> np.ndarray:
    """One BN254 G1 MSM over all devices of a multi-device context: sets[i] is the same point set on ctx.device(i); host
    scalars.  Returns the G1Jac image (12 words), equal to msm_g1 on one device."""
    arr = (C.c_void_p * len(sets))(*[p.handle for p in sets])
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(12, dtype=np.uint64)
    ctx._chk(ctx.L.dvn_msm_g1_multi(ctx.h, arr, len(sets), off, scalars.shape[0], _host(scalars), _host(out)))
    return out


class Comm:
    """RCCL communicator of the one-process-per-GPU deployment (dvn_comm_*): all-gather of fixed-size uint64 rows, host in / out.
    `Comm.unique_id()` on rank 0, its 128 bytes to every rank, then `Comm(ctx, id, rank, world)` on all of them (collective)."""

    @staticmethod
    def unique_id() -> bytes:
        L = load_library()
        buf = np.zeros(128, dtype=np.uint8)
        rc = L.dvn_comm_unique_id(C.c_void_p(buf.ctypes.data))
        if rc != 0:
            raise DvnError(rc, (L.dvn_last_error(None) or b"").decode())
        return buf.tobytes()

    def __init__(self, ctx: "Context", uid: bytes, rank: int, world: int):
        self.ctx, self.rank, self.world = ctx, rank, world
        buf = np.frombuffer(bytes(uid), dtype=np.uint8).copy()
        assert buf.size == 128
        h = C.c_void_p()
        ctx._chk(ctx.L.dvn_comm_init(ctx.h, C.c_void_p(buf.ctypes.data), rank, world, C.byref(h)))
        self.h = h

    def all_gather(self, local: np.ndarray) -> np.ndarray:
        local = np.ascontiguousarray(local, dtype=np.uint64).reshape(-1)
        out = np.zeros((self.world, local.size), dtype=np.uint64)
        self.ctx._chk(self.ctx.L.dvn_comm_all_gather(self.ctx.h, self.h, _host(local), local.size, _host(out)))
        return out

    def free(self):
        if self.h:
            self.ctx._chk(self.ctx.L.dvn_comm_free(self.ctx.h, self.h))
            self.h = None


def prove_committed_multi(ctx: "Context", keys, witness: np.ndarray, r_mont, s_mont) -> dict:
    """One committed (statetransition-shaped) BN254 proof over all devices of a multi-device context."""
    arr = (C.c_void_p * len(keys))(*[k.h for k in keys])
    witness = np.ascontiguousarray(witness, dtype=np.uint64)
    r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
    out = np.zeros(32, dtype=np.uint64); cm = np.zeros(8, dtype=np.uint64); pok = np.zeros(8, dtype=np.uint64); ch = np.zeros(4, dtype=np.uint64)
    ctx._chk(ctx.L.dvn_groth16_prove_committed_multi(ctx.h, arr, len(keys), _host(witness), _host(r_mont), _host(s_mont), _host(out), _host(cm), _host(pok), _host(ch)))
    return {"Ar": out[0:8].copy(), "Bs": out[8:24].copy(), "Krs": out[24:32].copy(), "Commitment": cm, "CommitmentPok": pok, "challenge": ch}


def gnark_pk_scan(data: bytes, curve: str = "bn254") -> dict:
    """Host only: the section sizes of a gnark proving-key image of `curve` (dvn_[<curve>_]gnark_pk_scan)."""
    L = load_library()
    info = (C.c_uint64 * 12)()
    fn = L.dvn_gnark_pk_scan if curve == "bn254" else getattr(L, f"dvn_{curve}_gnark_pk_scan")
    rc = fn(data, len(data), info)
    if rc != 0:
        raise DvnError(rc, L.dvn_last_error(None).decode())
    names = ("log_n", "raw", "n_wires", "A", "B", "Z", "K", "B2", "n_commitments", "basis0", "domain_flag", "bitmap_prefix")
    return dict(zip(names, (int(v) for v in info)))


class LoadedKey:
    """A proving key read from a gnark artifact file / image (dvn_[<curve>_]groth16_pk_load_file / _load_bytes): proofs come from
    the solver's output like ImportedKey's.  n_public and the committed private wires come from the constraint system gnark
    loads next to the key."""

    def __init__(self, ctx: "Context", source, n_public: int, committed_wires=(), sha256_hex: str | None = None, curve: str = "bn254"):
        a1 = {"bn254": 8, "bw6761": 24}.get(curve, 12)
        self.ctx, self._a1, self._a2 = ctx, a1, (a1 if curve == "bw6761" else 2 * a1)
        self._prefix = "dvn_" if curve == "bn254" else f"dvn_{curve}_"
        cw = np.ascontiguousarray(committed_wires, dtype=np.uint32)
        cwp = C.c_void_p(cw.ctypes.data) if cw.size else None
        h = C.c_void_p()
        if isinstance(source, (bytes, bytearray)):
            ctx._chk(getattr(ctx.L, self._prefix + "groth16_pk_load_bytes")(ctx.h, bytes(source), len(source), n_public, cwp, cw.size, C.byref(h)))
        else:
            ctx._chk(getattr(ctx.L, self._prefix + "groth16_pk_load_file")(ctx.h, str(source).encode(), sha256_hex.encode() if sha256_hex else None, n_public, cwp, cw.size, C.byref(h)))
        self.h = h
        ln = C.c_uint32(); sizes = (C.c_uint64 * 5)()
        ctx.L.dvn_groth16_pk_info(h, C.byref(ln), sizes)
        self.log_n = ln.value
        self.sizes = dict(zip(("A", "B", "K", "Z", "B2"), (int(v) for v in sizes)))

    prove_solved = None     # bound below to ImportedKey.prove_solved

    def commitment_bases(self):
        """(Basis, BasisExpSigma) Points handles owned by the key, or (None, None)."""
        b, s = C.c_void_p(), C.c_void_p()
        self.ctx._chk(self.ctx.L.dvn_groth16_pk_commitment_bases(self.h, C.byref(b), C.byref(s)))
        return (Points(self.ctx, b, 1) if b.value else None, Points(self.ctx, s, 1) if s.value else None)

    def last_stats(self) -> dict:
        return _stats_json(self.ctx, self.h)

    def free(self):
        if self.h:
            self.ctx._chk(self.ctx.L.dvn_groth16_pk_free(self.ctx.h, self.h))
            self.h = None

    def evict(self):
        """Drop the key from the context's sha256 cache and release it (every handle to it becomes invalid)."""
        if self.h:
            self.ctx._chk(self.ctx.L.dvn_groth16_pk_evict(self.ctx.h, self.h))
            self.h = None


class KeyArrays(C.Structure):
    """dvn_groth16_key_arrays (include/dvn_prover.h)."""
    _fields_ = [("log_n", C.c_uint32), ("n_wires", C.c_uint64),
                ("A", C.c_void_p), ("B", C.c_void_p), ("K", C.c_void_p), ("Z", C.c_void_p), ("B2", C.c_void_p),
                ("nA", C.c_uint64), ("nB", C.c_uint64), ("nK", C.c_uint64), ("nZ", C.c_uint64),
                ("idxA", C.c_void_p), ("idxB", C.c_void_p), ("idxK", C.c_void_p),
                ("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("delta_g1", C.c_void_p), ("beta_g2", C.c_void_p), ("delta_g2", C.c_void_p)]


class ImportedKey:
    """A gnark proving key handed over as arrays (dvn_groth16_pk_import); proofs come from the solver's output
    (dvn_groth16_prove_solved).  arrays: dict with A, B, K, Z (n,8), B2 (n,16) uint64; idxA, idxB, idxK uint32; consts (56,) uint64."""

    def __init__(self, ctx: "Context", log_n: int, n_wires: int, arrays: dict, curve: str = "bn254"):
        self.ctx = ctx
        self._prefix = "dvn_" if curve == "bn254" else f"dvn_{curve}_"
        self._a1 = {"bn254": 8, "bw6761": 24}.get(curve, 12)
        self._a2 = self._a1 if curve == "bw6761" else 2 * self._a1
        keep = {k: np.ascontiguousarray(arrays[k], dtype=np.uint64) for k in ("A", "B", "K", "Z", "B2", "consts")}
        keep.update({k: np.ascontiguousarray(arrays[k], dtype=np.uint32) for k in ("idxA", "idxB", "idxK")})
        c = keep["consts"]
        ptr = lambda a: a.ctypes.data
        ka = KeyArrays(log_n, n_wires, ptr(keep["A"]), ptr(keep["B"]), ptr(keep["K"]), ptr(keep["Z"]), ptr(keep["B2"]),
                       keep["A"].shape[0], keep["B"].shape[0], keep["K"].shape[0], keep["Z"].shape[0],
                       ptr(keep["idxA"]), ptr(keep["idxB"]), ptr(keep["idxK"]),
                       ptr(c), ptr(c) + 8 * self._a1, ptr(c) + 16 * self._a1, ptr(c) + 24 * self._a1, ptr(c) + 24 * self._a1 + 8 * self._a2)
        h = C.c_void_p()
        ctx._chk(getattr(ctx.L, self._prefix + "groth16_pk_import")(ctx.h, C.byref(ka), C.byref(h)))
        self.h = h

    def prove_solved(self, w: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray, r_mont, s_mont) -> dict:
        w, a, b, c = (np.ascontiguousarray(x, dtype=np.uint64) for x in (w, a, b, c))
        assert a.shape == b.shape == c.shape
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
        a1, a2 = self._a1, self._a2
        out = np.zeros(2 * a1 + a2, dtype=np.uint64)
        self.ctx._chk(getattr(self.ctx.L, self._prefix + "groth16_prove_solved")(self.ctx.h, self.h, _host(w), _host(a), _host(b), _host(c), a.shape[0],
                                                                                 _host(r_mont), _host(s_mont), _host(out)))
        return {"Ar": out[0:a1].copy(), "Bs": out[a1:a1 + a2].copy(), "Krs": out[a1 + a2:2 * a1 + a2].copy()}

    def prove_solved_shard(self, w: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray, rank: int, world: int) -> np.ndarray:
        """rank's share of prove_solved: the partial results of the five MSMs (4 G1 + 1 G2 Jacobian images; combine with prove_solved_combine)."""
        w, a, b, c = (np.ascontiguousarray(x, dtype=np.uint64) for x in (w, a, b, c))
        out = np.zeros(self.partial_words, dtype=np.uint64)
        self.ctx._chk(getattr(self.ctx.L, self._prefix + "groth16_prove_solved_shard")(self.ctx.h, self.h, _host(w), _host(a), _host(b), _host(c), a.shape[0], rank, world, _host(out)))
        return out

    @property
    def partial_words(self) -> int:
        return 6 * self._a1 + 3 * self._a2 // 2              # four G1 images of 3 coordinates, one G2 image

    def prove_solved_combine(self, partials: np.ndarray, r_mont, s_mont) -> dict:
        partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, self.partial_words)
        r_mont = np.ascontiguousarray(r_mont, dtype=np.uint64); s_mont = np.ascontiguousarray(s_mont, dtype=np.uint64)
        a1, a2 = self._a1, self._a2
        out = np.zeros(2 * a1 + a2, dtype=np.uint64)
        self.ctx._chk(getattr(self.ctx.L, self._prefix + "groth16_prove_combine")(self.h, _host(partials), partials.shape[0], _host(r_mont), _host(s_mont), _host(out)))
        return {"Ar": out[0:a1].copy(), "Bs": out[a1:a1 + a2].copy(), "Krs": out[a1 + a2:2 * a1 + a2].copy()}

    def free(self):
        if self.h:
            self.ctx._chk(self.ctx.L.dvn_groth16_pk_free(self.ctx.h, self.h))
            self.h = None


LoadedKey.prove_solved = ImportedKey.prove_solved
ImportedKey.last_stats = lambda self: _stats_json(self.ctx, self.h)


class Bls12377:
    """BLS12-377 entry points over an existing Context.  Points: (n,12) G1 / (n,24) G2 uint64."""
    PREFIX = "dvn_bls12377_"

    def _f(self, name):
        return getattr(self.L, self.PREFIX + name)

    def __init__(self, ctx: Context):
        self.ctx, self.L = ctx, ctx.L

    def upload_g1(self, xy: np.ndarray) -> Points:
        xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 12)
        p = C.c_void_p(); self.ctx._chk(self._f("points_upload_g1")(self.ctx.h, _host(xy), xy.shape[0], C.byref(p)))
        return Points(self.ctx, p, 1)

    def upload_g2(self, xy: np.ndarray) -> Points:
        xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 24)
        p = C.c_void_p(); self.ctx._chk(self._f("points_upload_g2")(self.ctx.h, _host(xy), xy.shape[0], C.byref(p)))
        return Points(self.ctx, p, 2)

    def synthetic_g1(self, seed: int, n: int) -> Points:
        p = C.c_void_p(); self.ctx._chk(self._f("points_synthetic_g1")(self.ctx.h, seed, n, C.byref(p)))
        return Points(self.ctx, p, 1)

    def download(self, pts: Points, off: int = 0, n: int | None = None) -> np.ndarray:
        n = len(pts) - off if n is None else n
        out = np.zeros((n, 12 if pts.curve == 1 else 24), dtype=np.uint64)
        self.ctx._chk(self._f("points_download")(self.ctx.h, pts.handle, off, n, _host(out)))
        return out

    def synthetic_scalars(self, seed: int, kind: int, n: int) -> int:
        d = self.ctx.malloc(n * 32)
        self.ctx._chk(self._f("scalars_synthetic")(self.ctx.h, seed, kind, n, C.c_void_p(d)))
        return d

    def msm_g1(self, pts: Points, scalars, off: int = 0, n: int | None = None) -> np.ndarray:
        sp, ns, fl = self.ctx._scalars(scalars, n)
        out = np.zeros(18, dtype=np.uint64)
        self.ctx._chk(self._f("msm_g1")(self.ctx.h, pts.handle, off, ns, sp, fl, _host(out)))
        return out

    def msm_g2(self, pts: Points, scalars, off: int = 0, n: int | None = None) -> np.ndarray:
        sp, ns, fl = self.ctx._scalars(scalars, n)
        out = np.zeros(36, dtype=np.uint64)
        self.ctx._chk(self._f("msm_g2")(self.ctx.h, pts.handle, off, ns, sp, fl, _host(out)))
        return out

    def domain(self, log_n: int) -> Domain:
        p = C.c_void_p(); self.ctx._chk(self._f("ntt_domain")(self.ctx.h, log_n, C.byref(p)))
        return Domain(self.ctx, p, log_n)

    def compute_h(self, dom: Domain, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
        """gnark computeH on host (len, 4) uint64 arrays; returns (n, 4), bit-reversed."""
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint64) for x in (a, b, c))
        out = np.zeros((dom.n, 4), dtype=np.uint64)
        self.ctx._chk(self._f("groth16_compute_h")(self.ctx.h, dom.handle, _host(a), _host(b), _host(c), a.shape[0], _host(out), 0))
        return out

    def pedersen_commit(self, basis: Points, basis_exp_sigma: Points, values: np.ndarray):
        """(commitment, pok) G1Affine images (12 uint64 each) of the BSB22 commitment hint on this curve."""
        values = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        c, k = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        self.ctx._chk(self._f("pedersen_commit")(self.ctx.h, basis.handle, basis_exp_sigma.handle, _host(values), values.shape[0], 0, _host(c), _host(k)))
        return c, k

    def ntt(self, dom: Domain, data, inverse: bool, decimation: int, coset: bool = False):
        if isinstance(data, (int, np.integer)):
            self.ctx._chk(self._f("ntt")(self.ctx.h, dom.handle, C.c_void_p(int(data)), int(inverse), decimation, int(coset), NTT_DATA_ON_DEVICE))
            return data
        assert data.dtype == np.uint64 and data.flags["C_CONTIGUOUS"] and data.shape == (dom.n, 4)
        self.ctx._chk(self._f("ntt")(self.ctx.h, dom.handle, _host(data), int(inverse), decimation, int(coset), 0))
        return data


class Bls12381(Bls12377):
    """BLS12-381 entry points (the blob-KZG curve): same element sizes as BLS12-377."""
    PREFIX = "dvn_bls12381_"


class Bw6761:
    """BW6-761 MSM entry points over an existing Context.  Points: (n,24) uint64 for either
    group; scalars (n,6) uint64."""

    def __init__(self, ctx: Context):
        self.ctx, self.L = ctx, ctx.L

    def upload(self, group: int, xy: np.ndarray) -> Points:
        assert xy.dtype == np.uint64 and xy.ndim == 2 and xy.shape[1] == 24 and xy.flags["C_CONTIGUOUS"]
        p = C.c_void_p(); self.ctx._chk(self.L.dvn_bw6761_points_upload(self.ctx.h, group, _host(xy), xy.shape[0], C.byref(p)))
        return Points(self.ctx, p, group)

    def synthetic(self, group: int, base_xy: np.ndarray, seed: int, n: int) -> Points:
        base_xy = np.ascontiguousarray(base_xy, dtype=np.uint64).reshape(24)
        p = C.c_void_p(); self.ctx._chk(self.L.dvn_bw6761_points_synthetic(self.ctx.h, group, _host(base_xy), seed, n, C.byref(p)))
        return Points(self.ctx, p, group)

    def download(self, pts: Points, off: int = 0, n: int | None = None) -> np.ndarray:
        n = len(pts) - off if n is None else n
        out = np.zeros((n, 24), dtype=np.uint64)
        self.ctx._chk(self.L.dvn_bw6761_points_download(self.ctx.h, pts.handle, off, n, _host(out)))
        return out

    def synthetic_scalars(self, seed: int, kind: int, n: int) -> int:
        d = self.ctx.malloc(n * 48)
        self.ctx._chk(self.L.dvn_bw6761_scalars_synthetic(self.ctx.h, seed, kind, n, C.c_void_p(d)))
        return d

    def msm(self, pts: Points, scalars, off: int = 0, n: int | None = None) -> np.ndarray:
        if isinstance(scalars, (int, np.integer)):
            assert n is not None
            sp, ns, fl = C.c_void_p(int(scalars)), n, MSM_SCALARS_ON_DEVICE
        else:
            scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 6)
            sp, ns, fl = _host(scalars), scalars.shape[0], 0
        out = np.zeros(36, dtype=np.uint64)
        self.ctx._chk(self.L.dvn_bw6761_msm(self.ctx.h, pts.handle, off, ns, sp, fl, _host(out)))
        return out

    def pedersen_commit(self, basis: Points, basis_exp_sigma: Points, values: np.ndarray):
        """(commitment, pok) G1Affine images (24 uint64 each): the aggregator circuit's BSB22 commitment hint."""
        values = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 6)
        c, k = np.zeros(24, dtype=np.uint64), np.zeros(24, dtype=np.uint64)
        self.ctx._chk(self.L.dvn_bw6761_pedersen_commit(self.ctx.h, basis.handle, basis_exp_sigma.handle, _host(values), values.shape[0], 0, _host(c), _host(k)))
        return c, k

    def domain(self, log_n: int) -> Domain:
        p = C.c_void_p(); self.ctx._chk(self.L.dvn_bw6761_ntt_domain(self.ctx.h, log_n, C.byref(p)))
        return Domain(self.ctx, p, log_n)

    def ntt(self, dom: Domain, data, inverse: bool, decimation: int, coset: bool = False):
        """data: (n, 6) uint64 host array (in place) or a device pointer."""
        if isinstance(data, (int, np.integer)):
            self.ctx._chk(self.L.dvn_bw6761_ntt(self.ctx.h, dom.handle, C.c_void_p(int(data)), int(inverse), decimation, int(coset), NTT_DATA_ON_DEVICE))
            return data
        assert data.dtype == np.uint64 and data.flags["C_CONTIGUOUS"] and data.shape == (dom.n, 6)
        self.ctx._chk(self.L.dvn_bw6761_ntt(self.ctx.h, dom.handle, _host(data), int(inverse), decimation, int(coset), 0))
        return data

    def compute_h(self, dom: Domain, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint64) for x in (a, b, c))
        out = np.zeros((dom.n, 6), dtype=np.uint64)
        self.ctx._chk(self.L.dvn_bw6761_groth16_compute_h(self.ctx.h, dom.handle, _host(a), _host(b), _host(c), a.shape[0], _host(out), 0))
        return out


class KzgSetup:
    """Device-resident EIP-4844 trusted setup + the blob operations of the reference's types.Blob
    (/root/reference/types/blobs.go:88-133), same names and byte formats:
      compute_commitment(blob) -> 48 bytes            Blob.ComputeCommitment
      compute_proof(blob, z)   -> (48 bytes, y: int)  Blob.ComputeProof (z: int; error if it does not fit the field)
      compute_cell_proofs(blob) -> list of 128 x 48 bytes   Blob.ComputeCellProofs
      evaluate_barycentric(blob, z) -> int            crypto/blobs/barycentric.go EvaluateBarycentricNative
    g1_lagrange / g1_monomial: 4096 x 48-byte compressed points in the order of config/kzg_trusted_setup.txt."""
    BLOB_BYTES = 131072

    def __init__(self, ctx: "Context", g1_lagrange: bytes, g1_monomial: bytes = None):
        if len(g1_lagrange) != 4096 * 48 or (g1_monomial is not None and len(g1_monomial) != 4096 * 48):
            raise ValueError("trusted setup bases must be 4096 x 48 bytes")
        self.ctx = ctx
        h = C.c_void_p()
        ctx._chk(ctx.L.dvn_kzg_setup_load(ctx.h, g1_lagrange, g1_monomial, C.byref(h)))
        self.h = h

    @classmethod
    def from_file(cls, ctx: "Context", path: str):
        """Text format of config/kzg_trusted_setup.txt: n1, n2, n1 G1 Lagrange, n2 G2 monomial, n1 G1 monomial (hex lines)."""
        t = open(path).read().split()
        n1, n2 = int(t[0]), int(t[1])
        lag = bytes.fromhex("".join(t[2:2 + n1]))
        mono = bytes.fromhex("".join(t[2 + n1 + n2:2 + 2 * n1 + n2])) if len(t) >= 2 + 2 * n1 + n2 else None
        return cls(ctx, lag, mono)

    def close(self):
        if self.h:
            self.ctx._chk(self.ctx.L.dvn_kzg_setup_free(self.ctx.h, self.h))
            self.h = None

    def _blob(self, blob: bytes) -> bytes:
        if len(blob) != self.BLOB_BYTES:
            raise ValueError("blob length is %d, want 131072" % len(blob))
        return bytes(blob)

    @staticmethod
    def _point(z: int) -> bytes:
        if z < 0 or z.bit_length() > 256:
            raise ValueError("point does not fit in 32 bytes")              # types/blobs.go:122-124
        return z.to_bytes(32, "big")

    def compute_commitment(self, blob: bytes) -> bytes:
        out = C.create_string_buffer(48)
        self.ctx._chk(self.ctx.L.dvn_kzg_blob_to_commitment(self.ctx.h, self.h, self._blob(blob), out))
        return out.raw

    def compute_proof(self, blob: bytes, z: int):
        proof, y = C.create_string_buffer(48), C.create_string_buffer(32)
        self.ctx._chk(self.ctx.L.dvn_kzg_compute_proof(self.ctx.h, self.h, self._blob(blob), self._point(z), proof, y))
        return proof.raw, int.from_bytes(y.raw, "big")

    def compute_cell_proofs(self, blob: bytes):
        out = C.create_string_buffer(128 * 48)
        self.ctx._chk(self.ctx.L.dvn_kzg_compute_cell_proofs(self.ctx.h, self.h, self._blob(blob), out))
        return [out.raw[48 * k:48 * k + 48] for k in range(128)]

    def compute_blob_proof(self, blob: bytes, commitment: bytes) -> bytes:
        """Blob.ComputeBlobProof (types/blobs.go:108-117): does not verify that the commitment belongs to the blob."""
        if len(commitment) != 48:
            raise ValueError("commitment must be 48 bytes")
        out = C.create_string_buffer(48)
        self.ctx._chk(self.ctx.L.dvn_kzg_compute_blob_proof(self.ctx.h, self.h, self._blob(blob), bytes(commitment), out))
        return out.raw

    def compute_commitment_and_proof(self, blob: bytes):
        """Blob.ComputeCommitmentAndProof (types/blobs.go:138-148): Version 0 sidecar material."""
        c = self.compute_commitment(blob)
        return c, self.compute_blob_proof(blob, c)

    def compute_commitment_and_cell_proofs(self, blob: bytes):
        """Blob.ComputeCommitmentAndCellProofs (types/blobs.go:152-162): Version 1 sidecar material."""
        return self.compute_commitment(blob), self.compute_cell_proofs(blob)

    def evaluate_barycentric(self, blob: bytes, z: int) -> int:
        y = C.create_string_buffer(32)
        self.ctx._chk(self.ctx.L.dvn_kzg_evaluate_barycentric(self.ctx.h, self.h, self._blob(blob), self._point(z), y))
        return int.from_bytes(y.raw, "big")
