"""Prove -> verify round trips, as the reference's own prover tests are, through a restatement of the
reference's IN-TREE Solidity verifier (/root/reference/config/statetransition_vkey.sol:393-520,654-744;
oracle/bn254_pairing_py.py) and of gnark's plain groth16.Verify (circuits/artifacts.go:595-613)."""
import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


@pytest.fixture(scope="module")
def pp():
    from oracle import bn254_pairing_py
    return bn254_pairing_py


def mont(o, x):
    return np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)


def calldata_words(o, e):
    """solidity/solidity.go:29-73: [A.x, A.y, B.x.A1, B.x.A0, B.y.A1, B.y.A0, C.x, C.y], commitments, pok."""
    a, b, c = e["Ar"], e["Bs"], e["Krs"]
    cm, pk = e["Commitment"] or (0, 0), e["CommitmentPok"] or (0, 0)
    return [a[0], a[1], b[0][1], b[0][0], b[1][1], b[1][0], c[0], c[1]], [cm[0], cm[1]], [pk[0], pk[1]]


def test_pairing_is_bilinear_on_the_reference_key_points(pp, opy, refconst):
    """Pins the pairing: non-degenerate, order r, bilinear -- on the generators and on points of the reference's verifying key
    (ALPHA, PEDERSEN_G of config/statetransition_vkey.sol:62-114)."""
    o = opy
    g1 = tuple(int(v) for v in refconst["g1_points"]["ALPHA"])
    q = refconst["g2_points"]["PEDERSEN_G"]
    g2 = ((int(q[0][0]), int(q[0][1])), (int(q[1][0]), int(q[1][1])))
    for p1, q2 in ((o.G1_GEN, o.G2_GEN), (g1, g2)):
        e = pp.pairing(q2, p1)
        assert not (e == pp.F12.one()) and e ** R == pp.F12.one()
        assert pp.pairing(o.g2_mul(q2, 11), o.g1_mul(p1, 7)) == e ** 77
        assert pp.pairing_check([(o.g1_mul(p1, 5), q2), (o.g1_neg(p1), o.g2_mul(q2, 5))])
        assert not pp.pairing_check([(o.g1_mul(p1, 5), q2), (o.g1_neg(p1), o.g2_mul(q2, 6))])
    assert pp.pairing_check([(None, o.G2_GEN), (o.G1_GEN, None)])          # EIP-197: infinity pairs contribute 1


def test_solidity_verifier_accepts_the_oracle_proof_and_rejects_tampering(pp, opy):
    """The restated verifier against the oracle's closed-form proof with one commitment (8 inputs + commitment wire, the
    statetransition shape), then every piece of calldata perturbed in turn."""
    o = opy
    cs = o.SyntheticR1CSCommitted(40, 9, 6, 7)
    e = o.groth16_committed_expected(cs, 3, 1234, 5678)
    proof, cm, pok = calldata_words(o, e)
    inputs = e["witness"][1:cs.n_inputs]
    assert len(inputs) == 8 and e["vk"]["PUBLIC_COMMITTED"] == [2]           # vkey.sol:665 hashes the third input
    assert pp.verify_proof_sol(e["vk"], proof, cm, pok, inputs, o, o.keccak256)
    bad = list(inputs); bad[5] = (bad[5] + 1) % R
    assert not pp.verify_proof_sol(e["vk"], proof, cm, pok, bad, o, o.keccak256)
    bad = list(inputs); bad[2] = (bad[2] + 1) % R                            # the hashed input: the challenge changes
    assert not pp.verify_proof_sol(e["vk"], proof, cm, pok, bad, o, o.keccak256)
    other = o.g1_mul(o.G1_GEN, 99)
    assert not pp.verify_proof_sol(e["vk"], proof, [other[0], other[1]], pok, inputs, o, o.keccak256)     # CommitmentInvalid
    assert not pp.verify_proof_sol(e["vk"], proof, cm, [other[0], other[1]], inputs, o, o.keccak256)
    p2 = list(proof); p2[6], p2[7] = other
    assert not pp.verify_proof_sol(e["vk"], p2, cm, pok, inputs, o, o.keccak256)                         # ProofInvalid
    assert not pp.verify_proof_sol(e["vk"], proof, cm, pok, [R] + list(inputs[1:]), o, o.keccak256)       # PublicInputNotInField


def _pt1(o, limbs):
    return o.g1_from_limbs([int(v) for v in limbs])


def _pt2(o, limbs):
    return o.g2_from_limbs([int(v) for v in limbs])


@pytest.mark.gpu
@pytest.mark.parametrize("n_constraints,n_committed", [(60, 6), (1 << 12, 300)])
def test_device_proof_with_commitment_passes_the_solidity_verifier(ctx, pkg, pp, opy, n_constraints, n_committed):
    """End to end as the reference's strongest test (statetransition_test.go:198-249): Setup -> Prove -> ABI calldata ->
    verifyProof.  The verifying key comes from the library's Setup and equals the oracle's bit for bit."""
    o = opy
    cs = pkg.R1CS(7, n_constraints, 9, n_committed=n_committed)
    key = pkg.Groth16Key(ctx, cs, 3)
    ref = o.SyntheticR1CSCommitted(n_constraints, 9, n_committed, 7)
    vkd = key.verifying_key(cs.n_public, committed=True)
    want = o.groth16_committed_expected(ref, 3, 0, 0)
    vk = {"ALPHA": _pt1(o, vkd["alpha"]), "BETA_NEG": o.g2_neg(_pt2(o, vkd["beta"])), "GAMMA_NEG": o.g2_neg(_pt2(o, vkd["gamma"])),
          "DELTA_NEG": o.g2_neg(_pt2(o, vkd["delta"])), "CONSTANT": _pt1(o, vkd["ic"][0]), "PUB": [_pt1(o, r) for r in vkd["ic"][1:]],
          "PEDERSEN_G": _pt2(o, vkd["pedersen_g"]), "PEDERSEN_GSIGMANEG": _pt2(o, vkd["pedersen_gsigmaneg"]), "PUBLIC_COMMITTED": [2]}
    assert vk == want["vk"]
    r, s = 0xABCDEF123456789, 0x1122334455
    got = key.prove_committed(cs.witness(), mont(o, r), mont(o, s))
    out = np.zeros(384, dtype=np.uint8)
    proof = np.concatenate([got["Ar"], got["Bs"], got["Krs"]])
    assert pkg.load_library().dvn_groth16_proof_abi_encode(proof.ctypes.data, got["Commitment"].ctypes.data, got["CommitmentPok"].ctypes.data, out.ctypes.data) == 0
    words = [int.from_bytes(bytes(out[32 * i:32 * i + 32]), "big") for i in range(12)]
    inputs = [o.from_mont(o.from_limbs(rw), R) for rw in cs.witness()[1:9]]
    assert pp.verify_proof_sol(vk, words[0:8], words[8:10], words[10:12], inputs, o, o.keccak256)
    bad = list(inputs); bad[0] = (bad[0] + 1) % R
    assert not pp.verify_proof_sol(vk, words[0:8], words[8:10], words[10:12], bad, o, o.keccak256)
    key.free(); cs.free()


@pytest.mark.gpu
def test_device_plain_proof_passes_groth16_verify(ctx, pkg, pp, opy):
    """No commitment: gnark's groth16.Verify equation with the library's verifying key (CircuitRuntime.ProveAndVerify shape)."""
    o = opy
    cs = pkg.R1CS(7, 500, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    vkd = key.verifying_key(cs.n_public)
    vk = {"ALPHA": _pt1(o, vkd["alpha"]), "BETA_NEG": o.g2_neg(_pt2(o, vkd["beta"])), "GAMMA_NEG": o.g2_neg(_pt2(o, vkd["gamma"])),
          "DELTA_NEG": o.g2_neg(_pt2(o, vkd["delta"])), "CONSTANT": _pt1(o, vkd["ic"][0]), "PUB": [_pt1(o, r) for r in vkd["ic"][1:]]}
    got = key.prove(cs.witness(), mont(o, 424242), mont(o, 777))
    inputs = [o.from_mont(o.from_limbs(rw), R) for rw in cs.witness()[1:9]]
    a, b, c = _pt1(o, got["Ar"]), _pt2(o, got["Bs"]), _pt1(o, got["Krs"])
    assert pp.verify_groth16(vk, a, b, c, inputs, o)
    assert not pp.verify_groth16(vk, a, b, c, inputs[:-1] + [(inputs[-1] + 1) % R], o)
    key.free(); cs.free()
