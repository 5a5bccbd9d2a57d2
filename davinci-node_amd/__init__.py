"""davinci-node_amd: MI355X-native prover backend for vocdoni/davinci-node (BN254 Groth16 hot path).

The product is the C-ABI shared library ``csrc/libdvn_prover.so`` (see ``include/dvn_prover.h``); the Go shim a
maintainer adds next to ``prover/prover_gpu.go`` binds it through cgo.  This Python package is
the ctypes harness over the same C-ABI that tests and bench.py use -- no arithmetic lives in Python, and there
is NO CPU fallback: if the HIP library is missing or no GPU is visible, calls raise.

The directory name contains a dash, so import it through ``__graft_entry__.load_package()`` (or
``importlib``) under the module name ``davinci_node_amd``.
"""
from . import sharding  # noqa: F401
from .binding import (  # noqa: F401
    DvnError, Context, Points, Domain, lib_path, load_library, build_library,
    DIF, DIT, R1CS, Groth16Key, Bls12377, Bls12381, Bw6761, KzgSetup, ImportedKey, LoadedKey, prove_multi, prove_committed_multi, msm_g1_multi, Comm, gnark_pk_scan,
)
