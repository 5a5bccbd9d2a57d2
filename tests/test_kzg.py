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


# ---------------------------------------------------------------- part 2: the HIP path (-m gpu)
@pytest.fixture(scope="module")
def dev(ctx, pkg):
    s = pkg.KzgSetup.from_file(ctx, os.path.join(GOLDEN, "kzg_trusted_setup.txt"))
    yield s
    s.close()


@pytest.mark.gpu
def test_gpu_setup_decompression_matches_reference_bases(dev, ts, K, ctx, pkg):
    """sum_i L_i = [1] and sum_i w_i L_bitrev(i) = [tau] = monomial[1] through the device MSM over the device-decompressed
    setup: pins decompression, the bit-reversed placement and the BLS12-381 MSM against the reference's own file."""
    ones = bytes(31) + b"\x01"
    assert dev.compute_commitment(ones * 4096) == K.g1_compress(K.G1_GEN)
    blob = b"".join(w.to_bytes(32, "big") for w in K.domain_brp(4096))
    assert dev.compute_commitment(blob) == ts.g1_monomial_bytes[1]
    # f = X^5 in evaluation form -> [tau^5]
    blob5 = b"".join(pow(w, 5, K.R).to_bytes(32, "big") for w in K.domain_brp(4096))
    assert dev.compute_commitment(blob5) == ts.g1_monomial_bytes[5]
    # every element r - 1 (all 4096 entries of every table row land in ONE bucket: the heavy-bucket path): -[1]G1
    neg_g = (K.G1_GEN[0], K.P - K.G1_GEN[1])
    assert dev.compute_commitment((K.R - 1).to_bytes(32, "big") * 4096) == K.g1_compress(neg_g)
    # the constant polynomial has zero quotients everywhere: all 128 cell proofs and the opening proof are the point at infinity
    inf = bytes([0xC0]) + bytes(47)
    assert dev.compute_cell_proofs(ones * 4096) == [inf] * 128
    assert dev.compute_proof(ones * 4096, 5) == (inf, 1)


@pytest.mark.gpu
def test_gpu_commitment(dev, ts, K, blobs):
    for blob in (blobs[0], blobs[1], small_blob(K, 7), small_blob(K, 1, 20)):
        assert dev.compute_commitment(blob) == K.blob_to_commitment(ts, blob)
    assert dev.compute_commitment(bytes(131072)) == bytes([0xC0]) + bytes(47)


@pytest.mark.gpu
def test_gpu_proof_and_claim(dev, ts, K, blobs):
    omega = K.domain_brp(4096)
    cases = [(blobs[0], 12345 * 7), (blobs[1], K.R - 1), (small_blob(K, 3), 3 * 12345), (blobs[1], omega[5]), (small_blob(K, 9), omega[4095]),
             (blobs[0], 0), (blobs[0], 1)]                                   # z = 1 = w_0 is in the domain
    for blob, z in cases:
        proof, y = dev.compute_proof(blob, z)
        want_proof, want_y = K.compute_proof(ts, blob, z)
        assert y == want_y and proof == want_proof, "z = %x" % z
        assert dev.evaluate_barycentric(blob, z) == K.evaluate_barycentric(blob, z) == y   # barycentric.go:24-170


@pytest.mark.gpu
def test_gpu_cell_proofs(dev, ts, K, blobs):
    blob = blobs[0]
    got = dev.compute_cell_proofs(blob)
    assert len(got) == 128 and len(set(got)) == 128
    coeffs = K.blob_coefficients(blob)
    for k in (0, 1, 64, 127):                                                # individual cells against the oracle
        assert got[k] == K.compute_cell_proof(ts, blob, k, coeffs), "cell %d" % k
    # all 128 at once: sum_k r_k proof_k == commitment to sum_k r_k q_k over the monomial basis
    rng = random.Random(11)
    rk = [rng.randrange(1 << 64) for _ in range(128)]
    comb = [0] * (4096 - 64)
    for k in range(128):
        q = K.cell_quotient_coefficients(coeffs, k)
        for j, v in enumerate(q):
            comb[j] = (comb[j] + rk[k] * v) % K.R
    want = K.g1_msm(ts.g1_monomial[:4096 - 64], comb)
    assert K.g1_msm([K.g1_decompress(p) for p in got], rk) == want
    # a sparse blob (most coefficients of every quotient equal): zero blob -> all proofs at infinity
    assert dev.compute_cell_proofs(bytes(131072)) == [bytes([0xC0]) + bytes(47)] * 128
    small = small_blob(K, 5)
    got2 = dev.compute_cell_proofs(small)
    c2 = K.blob_coefficients(small)
    assert got2[77] == K.compute_cell_proof(ts, small, 77, c2)


@pytest.mark.gpu
def test_gpu_kzg_errors(dev, ctx, pkg, K):
    bad = bytearray(131072); bad[32:64] = K.R.to_bytes(32, "big")
    with pytest.raises(pkg.DvnError):
        dev.compute_commitment(bytes(bad))
    with pytest.raises(pkg.DvnError):
        dev.compute_proof(bytes(131072), K.R)                                # non-canonical evaluation point
    with pytest.raises(ValueError):
        dev.compute_proof(bytes(131072), 1 << 256)                           # types/blobs_test.go:10 TestComputeProofRejectsOversizedPoint
    with pytest.raises(ValueError):
        dev.compute_commitment(bytes(100))
    lag = bytearray(4096 * 48)
    with pytest.raises(pkg.DvnError):
        pkg.KzgSetup(ctx, bytes(lag))                                        # not a compressed encoding
    lag2 = open(os.path.join(GOLDEN, "kzg_trusted_setup.txt")).read().split()
    pts = bytearray(bytes.fromhex("".join(lag2[2:2 + 4096])))
    pts[48 * 9 + 47] ^= 1                                                    # x no longer on the curve (or another point: the sums change)
    try:
        s = pkg.KzgSetup(ctx, bytes(pts))
        s.close()
    except pkg.DvnError as e:
        assert "curve" in str(e)
    no_mono = pkg.KzgSetup(ctx, bytes.fromhex("".join(lag2[2:2 + 4096])))
    with pytest.raises(pkg.DvnError):
        no_mono.compute_cell_proofs(bytes(131072))
    no_mono.close()


def test_sha256_matches_hashlib(pkg):
    import ctypes as C
    import hashlib
    L = pkg.load_library()
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000, 131072 + 80):
        m = bytes((5 * i + n) & 0xFF for i in range(n))
        out = C.create_string_buffer(32)
        assert L.dvn_sha256(m, n, out) == 0 and out.raw == hashlib.sha256(m).digest()


@pytest.mark.gpu
def test_gpu_blob_proof(dev, ts, K, blobs):
    """Blob.ComputeBlobProof / ComputeCommitmentAndProof (types/blobs.go:108-148)."""
    for blob in (blobs[0], small_blob(K, 7)):
        c, p = dev.compute_commitment_and_proof(blob)
        assert c == K.blob_to_commitment(ts, blob) and p == K.compute_blob_proof(ts, blob, c)
    other = K.g1_compress(K.G1_GEN)                        # not the blob's commitment: still the opening at the derived point
    assert dev.compute_blob_proof(blobs[1], other) == K.compute_blob_proof(ts, blobs[1], other)
    c, cells = dev.compute_commitment_and_cell_proofs(blobs[1])
    assert c == K.blob_to_commitment(ts, blobs[1]) and len(cells) == 128


@pytest.mark.gpu
def test_gpu_kzg_concurrent_callers(dev, ctx, pkg, K, ts, blobs, orc):
    """cgo calls arrive on arbitrary OS threads: blob operations, BN254 MSMs and a second setup on the same
    context interleave from four threads; every result must still be bit-exact."""
    import threading
    want_c = [dev.compute_commitment(b) for b in blobs]
    want_p = [dev.compute_proof(b, 777 + i) for i, b in enumerate(blobs)]
    want_cells = dev.compute_cell_proofs(blobs[0])
    n = 1 << 12
    pts_host = orc.g1_synthetic_points(0xDA71C1, n); scal = orc.scalars(1, 0, n)
    pts = ctx.upload_g1(pts_host)
    want_msm = orc.msm_g1(pts_host, scal)
    errs = []

    def worker(tid):
        try:
            for it in range(4):
                k = (tid + it) % 2
                if tid == 0:
                    assert dev.compute_cell_proofs(blobs[0]) == want_cells
                elif tid == 1:
                    assert (orc.g1_jac_to_affine(ctx.msm_g1(pts, scal)) == want_msm).all()
                else:
                    assert dev.compute_commitment(blobs[k]) == want_c[k]
                    assert dev.compute_proof(blobs[k], 777 + k) == want_p[k]
        except Exception as e:                                  # noqa: BLE001
            errs.append((tid, repr(e)))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    pts.free()
    assert not errs, errs


# ---------------------------------------------------------------- part 3: proofs VERIFIED by the setup's G2 side (pairing)
@pytest.fixture(scope="module")
def BP():
    from oracle import bls12381_pairing_py
    return bls12381_pairing_py


@pytest.fixture(scope="module")
def g2pts(BP):
    lines = open(os.path.join(GOLDEN, "kzg_trusted_setup.txt")).read().split()
    raw = [bytes.fromhex(h) for h in lines[2 + 4096:2 + 4096 + 65]]
    return {0: BP.g2_decompress(raw[0]), 1: BP.g2_decompress(raw[1]), 64: BP.g2_decompress(raw[64])}


def kzg_verify(K, BP, g2pts, commitment48, z, y, proof48):
    """EIP-4844 verify_kzg_proof_impl: e(pi, [tau]_2 - [z]_2) * e(-(C - [y]_1), [1]_2) == 1, G2 points from the reference's file."""
    c, pi = K.g1_decompress(commitment48), K.g1_decompress(proof48)
    lhs2 = BP.g2_add(g2pts[1], BP.g2_neg(BP.g2_mul(g2pts[0], z)))
    cy = K.g1_add(c, K.g1_mul(K.G1_GEN, (K.R - y) % K.R))
    neg = None if cy is None else (cy[0], (-cy[1]) % K.P)
    return BP.pairing_check([(pi, lhs2), (neg, g2pts[0])])


def cell_verify(K, BP, ts, g2pts, commitment48, coeffs, k, proof48):
    """EIP-7594 cell proof equation for one cell: e(pi_k, [tau^64]_2 - [c_k]_2) * e(-(C - [I_k(tau)]_1), [1]_2) == 1 with
    I_k = f mod (X^64 - c_k) taken straight from the coefficients (no quotient involved)."""
    ck = pow(K.cell_coset_shift(k), 64, K.R)
    ik = [sum(coeffs[j + 64 * t] * pow(ck, t, K.R) for t in range(64)) % K.R for j in range(64)]
    c, pi = K.g1_decompress(commitment48), K.g1_decompress(proof48)
    ci = K.g1_add(c, K.g1_msm(ts.g1_monomial[:64], [(K.R - v) % K.R for v in ik], c=4))
    neg = None if ci is None else (ci[0], (-ci[1]) % K.P)
    lhs2 = BP.g2_add(g2pts[64], BP.g2_neg(BP.g2_mul(g2pts[0], ck)))
    return BP.pairing_check([(pi, lhs2), (neg, g2pts[0])])


def test_pairing_pinned_by_the_two_sides_of_the_setup(K, BP, ts, g2pts):
    """The file's G1 and G2 powers share tau: e([tau^j]_1, [1]_2) == e([1]_1, [tau^j]_2) for j = 1, 64 -- pins the pairing (and
    both decompressions) on reference data; plus bilinearity, order and non-degeneracy."""
    g1, g2 = K.G1_GEN, g2pts[0]
    e = BP.pairing(g2, g1)
    assert not (e == BP.F12.one()) and e ** K.R == BP.F12.one()
    assert BP.pairing(BP.g2_mul(g2, 5), K.g1_mul(g1, 3)) == e ** 15
    ng1 = (g1[0], (-g1[1]) % K.P)
    for j in (1, 64):
        assert BP.pairing_check([(ts.g1_monomial[j], g2), (ng1, g2pts[j])])
    assert not BP.pairing_check([(ts.g1_monomial[2], g2), (ng1, g2pts[1])])


def test_oracle_proofs_verify_under_the_setup(K, BP, ts, g2pts, blobs):
    blob = blobs[0]
    c = K.blob_to_commitment(ts, blob)
    for z in (987654321, K.domain_brp(4096)[77]):
        proof, y = K.compute_proof(ts, blob, z)
        assert kzg_verify(K, BP, g2pts, c, z, y, proof)
        assert not kzg_verify(K, BP, g2pts, c, z, (y + 1) % K.R, proof)
    coeffs = K.blob_coefficients(blob)
    assert cell_verify(K, BP, ts, g2pts, c, coeffs, 9, K.compute_cell_proof(ts, blob, 9, coeffs))
    assert not cell_verify(K, BP, ts, g2pts, c, coeffs, 10, K.compute_cell_proof(ts, blob, 9, coeffs))


@pytest.mark.gpu
def test_gpu_proofs_verify_under_the_setup(dev, K, BP, ts, g2pts, blobs):
    """The HIP path's commitment / opening / blob / cell proofs accepted by the pairing equations over the reference's own G2
    points -- no oracle arithmetic between the device output and the check except the verifier's."""
    blob = blobs[1]
    c = dev.compute_commitment(blob)
    for z in (0xDEADBEEF12345, K.domain_brp(4096)[4000]):
        proof, y = dev.compute_proof(blob, z)
        assert kzg_verify(K, BP, g2pts, c, z, y, proof)
    bp_ = dev.compute_blob_proof(blob, c)
    zc = K.compute_challenge(blob, c)
    assert kzg_verify(K, BP, g2pts, c, zc, dev.evaluate_barycentric(blob, zc), bp_)
    cells = dev.compute_cell_proofs(blob)
    coeffs = K.blob_coefficients(blob)
    for k in (0, 37, 127):
        assert cell_verify(K, BP, ts, g2pts, c, coeffs, k, cells[k])
    assert not cell_verify(K, BP, ts, g2pts, c, coeffs, 1, cells[2])


@pytest.mark.gpu
def test_gpu_bls12381_g2_msm_pinned_by_the_setup(ctx, pkg, K, BP, ts):
    """The device's BLS12-381 G2 MSM over the file's 65 G2 powers: [p(tau)]_2 for a random degree-64 p must pair like the
    [p(tau)]_1 the G1 monomial basis gives -- e([p(tau)]_1, [1]_2) == e([1]_1, [p(tau)]_2)."""
    from oracle import curves_py
    S = curves_py.BLS12_381
    lines = open(os.path.join(GOLDEN, "kzg_trusted_setup.txt")).read().split()
    g2 = [BP.g2_decompress(bytes.fromhex(h)) for h in lines[2 + 4096:2 + 4096 + 65]]
    rng = random.Random(21)
    coef = [rng.randrange(K.R) for _ in range(65)]
    bls = pkg.Bls12381(ctx)
    pts = bls.upload_g2(np.array([S.g2_to_limbs(p) for p in g2], dtype=np.uint64))
    out = bls.msm_g2(pts, np.array([S.fr_to_limbs(c) for c in coef], dtype=np.uint64))
    assert out[24:36].any()                                                   # Z != 0
    x = (S.fp_from_mont(int(sum(int(out[i]) << (64 * i) for i in range(6)))), S.fp_from_mont(int(sum(int(out[6 + i]) << (64 * i) for i in range(6)))))
    y = (S.fp_from_mont(int(sum(int(out[12 + i]) << (64 * i) for i in range(6)))), S.fp_from_mont(int(sum(int(out[18 + i]) << (64 * i) for i in range(6)))))
    p1 = K.g1_msm(ts.g1_monomial[:65], coef, c=4)
    ng1 = (K.G1_GEN[0], (-K.G1_GEN[1]) % K.P)
    assert BP.pairing_check([(p1, g2[0]), (ng1, (x, y))])
    pts.free()
