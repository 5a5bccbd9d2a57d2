"""EIP-4844 / EIP-7594 blob KZG (/root/reference/types/blobs.go:88-133, crypto/blobs/blob.go:40-90).

Part 1 (CPU): the oracle (oracle/kzg_py.py) PINNED by data the reference itself holds -- the embedded trusted setup
(two bases of the same tau), the omega table of crypto/blobs/omega.go, the root-of-unity literal of barycentric.go:52 and
the reference's two test blobs.  Part 2 (-m gpu): the HIP path (dvn_kzg_*) bit-exact against that oracle and against the
same reference data directly."""
import json
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def K():
    from oracle import kzg_py
    return kzg_py


@pytest.fixture(scope="module")
def ts(K):
    return K.TrustedSetup(os.path.join(GOLDEN, "kzg_trusted_setup.txt"))


@pytest.fixture(scope="module")
def blobs():
    return [bytes.fromhex(open(os.path.join(GOLDEN, "kzg_blobdata%d.txt" % k)).read().strip()) for k in (1, 2)]


@pytest.fixture(scope="module")
def omega_fixture():
    return json.load(open(os.path.join(GOLDEN, "bls12381_omega_brp_4096.json")))


def small_blob(K, seed, count=50):
    """The blob shape of the reference's generateValidKZGData (crypto/blobs/testdata.go:103-113): `count` small values."""
    b = bytearray(32 * K.N)
    for i in range(count):
        b[32 * i:32 * i + 32] = (seed + i).to_bytes(32, "big")
    return bytes(b)


# ---------------------------------------------------------------- part 1: oracle pinned by reference data (CPU)
def test_roots_match_reference_tables(K, omega_fixture):
    from oracle import curves_py
    assert int(omega_fixture["root_of_unity_2^32"]) == K.ROOT_2_32 == curves_py.BLS12_381.root      # barycentric.go:52
    assert [int(h, 16) for h in omega_fixture["omega_brp"]] == K.domain_brp(4096)                  # omega.go:4-4101
    # the generic NTT oracle the HIP NTT is checked against: DIF of the unit impulse at index 1 lists w^bitrev(i)
    from oracle import bn254_py as b
    d = curves_py.BLS12_381.domain(4096)
    x = [0] * 4096; x[1] = 1
    d.fft(x, b.DIF)
    assert x == [int(h, 16) for h in omega_fixture["omega_brp"]]


def test_trusted_setup_is_consistent(K, ts):
    assert all(K.g1_on_curve(p) for p in ts.g1_lagrange) and all(K.g1_on_curve(p) for p in ts.g1_monomial)
    assert ts.g1_monomial[0] == K.G1_GEN
    assert K.g1_compress(ts.g1_monomial[5]) == ts.g1_monomial_bytes[5] and K.g1_compress(ts.g1_lagrange[77]) == ts.g1_lagrange_bytes[77]
    acc = None
    for p in ts.g1_lagrange:                              # sum_i L_i(tau) = 1
        acc = K.jac_add(acc, K.to_jac(p))
    assert K.to_affine(acc) == K.G1_GEN
    w = K.root_of_unity(4096)
    nat = [pow(w, i, K.R) for i in range(4096)]
    assert K.g1_msm(ts.g1_lagrange, nat) == ts.g1_monomial[1]            # sum_i w^i L_i(tau) = tau: L_i <-> w^i, natural order
    assert K.g1_msm(ts.g1_lagrange_brp, K.domain_brp(4096)) == ts.g1_monomial[1]


def test_commitment_lagrange_equals_monomial(K, ts, blobs):
    """blob_to_commitment over the Lagrange basis (bit-reversed) == the commitment to the interpolated coefficients over
    the monomial basis: ties MSM, domain order and inverse transform to the reference's setup file."""
    for blob in (blobs[0], small_blob(K, 7)):
        coeffs = K.blob_coefficients(blob)
        assert K.blob_to_commitment(ts, blob) == K.g1_compress(K.g1_msm(ts.g1_monomial, coeffs))
    assert K.blob_to_commitment(ts, bytes(32 * 4096)) == bytes([0xC0]) + bytes(47)     # zero blob -> infinity


def horner(K, coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % K.R
    return acc


def test_proof_and_claim(K, ts, blobs):
    blob = blobs[1]
    coeffs = K.blob_coefficients(blob)
    omega = K.domain_brp(4096)
    for z in (12345 * 7, omega[5]):                      # outside and inside the domain
        proof, y = K.compute_proof(ts, blob, z)
        assert y == horner(K, coeffs, z) == K.evaluate_barycentric(blob, z)
        # q(X) = (f(X) - y)/(X - z) by synthetic division, committed over the monomial basis
        q = [0] * 4095
        carry = 0
        for j in range(4095, 0, -1):
            carry = (coeffs[j] + carry * z) % K.R
            q[j - 1] = carry
        assert (coeffs[0] + carry * z) % K.R == y
        assert proof == K.g1_compress(K.g1_msm(ts.g1_monomial[:4095], q))
    assert K.evaluate_barycentric(blob, omega[9]) == K.blob_to_field_elements(blob)[9]   # barycentric.go:99-107


def test_cell_quotient(K, blobs):
    coeffs = K.blob_coefficients(blobs[0])
    rng = random.Random(5)
    for k in (0, 1, 77, 127):
        q = K.cell_quotient_coefficients(coeffs, k)
        h = K.cell_coset_shift(k)
        c = pow(h, 64, K.R)
        rem = [(coeffs[j] + c * q[j]) % K.R for j in range(64)]          # f = q (X^64 - c) + rem
        s = rng.randrange(K.R)
        assert horner(K, coeffs, s) == (horner(K, q, s) * (pow(s, 64, K.R) - c) + horner(K, rem, s)) % K.R
        x = h * pow(K.root_of_unity(64), 3, K.R) % K.R                   # a point of the cell's coset: the quotient vanishes there
        assert horner(K, coeffs, x) == horner(K, rem, x)
    # cosets of the 128 cells tile the 8192-point extended domain
    assert len({pow(K.cell_coset_shift(k), 64, K.R) for k in range(128)}) == 128


def test_blob_rejects_non_canonical(K):
    bad = bytearray(32 * 4096); bad[0:32] = K.R.to_bytes(32, "big")
    with pytest.raises(ValueError):
        K.blob_to_field_elements(bytes(bad))
    with pytest.raises(ValueError):
        K.blob_to_field_elements(bytes(100))
