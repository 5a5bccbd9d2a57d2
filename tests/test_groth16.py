"""Groth16 orchestration: the library's Setup + Prove on the synthetic statetransition-shaped
instance against the trapdoor identity -- the proof must equal [alpha + sum w_i A_i(tau) + r delta]G1 etc. bit for bit
for the injected (r, s).  The checker (oracle) uses no MSM and no NTT."""
import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def mont(o, x):
    return np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)


def test_synthetic_r1cs_witness_matches_oracle(pkg, opy, orc):
    """Host-only part of the library (no GPU needed): the seeded instance and its satisfying assignment."""
    cs = pkg.R1CS(11, 13, 3)
    assert (cs.n_wires, cs.n_public, cs.n_constraints) == (3 + 4 + 13, 3, 13)
    ref = opy.SyntheticR1CS(13, 3, 11)
    assert [opy.from_mont(opy.from_limbs(r), R) for r in cs.witness()] == ref.witness
    cs.free()
    cs = pkg.R1CS(5, 3000, 9)
    wit, _ = orc.groth16_synthetic_expected(5, 3000, 9, 3, mont(opy, 1), mont(opy, 2))
    assert (cs.witness() == wit).all()
    cs.free()


@pytest.mark.gpu
def test_groth16_small_instance_vs_python_oracle(ctx, pkg, opy):
    o = opy
    cs = pkg.R1CS(11, 13, 3)
    key = pkg.Groth16Key(ctx, cs, 3)
    ref = o.SyntheticR1CS(13, 3, 11)
    pk, trap = o.groth16_setup(ref, 3)
    r, s = 1234567, 7654321
    want = o.groth16_prove(ref, pk, r, s)                       # MSM/NTT-based python prover
    closed = o.groth16_expected_from_trapdoor(ref, trap, pk["domain"], r, s)
    assert want["Ar"] == closed["Ar"] and want["Bs"] == closed["Bs"] and want["Krs"] == closed["Krs"]
    got = key.prove(cs.witness(), mont(o, r), mont(o, s))
    assert list(got["Ar"]) == o.g1_to_limbs(want["Ar"])
    assert list(got["Bs"]) == o.g2_to_limbs(want["Bs"])
    assert list(got["Krs"]) == o.g1_to_limbs(want["Krs"])
    # key shape: gnark drops points at infinity from A and B (InfinityA/B)
    nzA = sum(1 for v in trap["At"] if v); nzB = sum(1 for v in trap["Bt"] if v)
    assert key.sizes == {"A": nzA, "B": nzB, "K": ref.n_wires - 3, "Z": 16, "B2": nzB}
    key.free(); cs.free()


@pytest.mark.gpu
# the last case is the statetransition-shaped instance of BASELINE config 5 (bench extras): n = 2^20, 8 public inputs + constant
@pytest.mark.parametrize("n_constraints,n_public", [(100, 1), (1 << 12, 9), ((1 << 14) - 5, 9), (1 << 16, 9), (1 << 20, 9)])
def test_groth16_trapdoor_identity(ctx, pkg, opy, orc, n_constraints, n_public):
    cs = pkg.R1CS(7, n_constraints, n_public)
    key = pkg.Groth16Key(ctx, cs, 3)
    for r, s in ((1234567, 7654321), (0, 0), (R - 1, 5)):
        wit, want = orc.groth16_synthetic_expected(7, n_constraints, n_public, 3, mont(opy, r), mont(opy, s))
        got = key.prove(cs.witness(), mont(opy, r), mont(opy, s))
        for k in ("Ar", "Bs", "Krs"):
            assert (got[k] == want[k]).all(), (k, r, s)
    # device-resident witness (HBM hand-off) gives the same proof
    d = ctx.to_device(cs.witness())
    got2 = key.prove(d, mont(opy, 1234567), mont(opy, 7654321))
    _, want = orc.groth16_synthetic_expected(7, n_constraints, n_public, 3, mont(opy, 1234567), mont(opy, 7654321))
    assert all((got2[k] == want[k]).all() for k in want)
    ctx.free(d); key.free(); cs.free()


@pytest.mark.gpu
def test_groth16_wrong_witness_changes_proof(ctx, pkg, opy, orc):
    """Sanity of the checker itself: a corrupted assignment must not reproduce the trapdoor proof."""
    cs = pkg.R1CS(7, 500, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    w = cs.witness(); w[100] = w[101]
    got = key.prove(w, mont(opy, 1), mont(opy, 2))
    _, want = orc.groth16_synthetic_expected(7, 500, 9, 3, mont(opy, 1), mont(opy, 2))
    assert not (got["Krs"] == want["Krs"]).all()
    key.free(); cs.free()


def test_proof_abi_encoding_matches_reference_layout(pkg, opy, orc):
    """solidity/solidity.go:29-73,85-116: [Ar.X, Ar.Y, Bs.X.A1, Bs.X.A0, Bs.Y.A1, Bs.Y.A0, Krs.X, Krs.Y] ++ commitments ++ pok."""
    import ctypes as C
    o = opy
    L = pkg.load_library()
    ar = o.g1_mul(o.G1_GEN, 12345); krs = o.g1_mul(o.G1_GEN, 99); bs = o.g2_mul(o.G2_GEN, 777); com = o.g1_mul(o.G1_GEN, 5)
    proof = np.array(o.g1_to_limbs(ar) + o.g2_to_limbs(bs) + o.g1_to_limbs(krs), dtype=np.uint64)
    cm = np.array(o.g1_to_limbs(com), dtype=np.uint64)
    out = np.zeros(384, dtype=np.uint8)
    assert L.dvn_groth16_proof_abi_encode(proof.ctypes.data, cm.ctypes.data, None, out.ctypes.data) == 0
    words = [int.from_bytes(bytes(out[32 * i:32 * i + 32]), "big") for i in range(12)]
    assert words == [ar[0], ar[1], bs[0][1], bs[0][0], bs[1][1], bs[1][0], krs[0], krs[1], com[0], com[1], 0, 0]


@pytest.mark.gpu
def test_pedersen_commit_and_pok(ctx, opy, orc):
    """Row a14: commitment = MSM(Basis, v), pok = MSM(BasisExpSigma, v) with BasisExpSigma = sigma * Basis, hence
    pok == sigma * commitment (what the verifier's pairing check e(C, -sigma G2) e(pok, G2) = 1 relies on)."""
    o = opy
    n, sigma = 700, 0x1234567890ABCDEF1234567
    basis = orc.g1_synthetic_points(81, n)
    bes = np.array([orc.g1_mul(basis[i], sigma) for i in range(n)], dtype=np.uint64)
    v = orc.scalars(9, 1, n)
    pb, ps = ctx.upload_g1(basis), ctx.upload_g1(bes)
    c, k = ctx.pedersen_commit(pb, ps, v)
    assert (c == orc.msm_g1(basis, v)).all() and (k == orc.msm_g1(bes, v)).all()
    assert (k == orc.g1_mul(c, sigma)).all()
    pb.free(); ps.free()


@pytest.mark.gpu
def test_groth16_sharded_by_rows_recombines(ctx, pkg, opy, orc):
    """BASELINE config 5: the partial results of any number of row-sharding ranks (played one after the other by this
    process) add up to the same proof as the unsharded prover and as the trapdoor identity."""
    n_c = (1 << 16) - 3
    cs = pkg.R1CS(7, n_c, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    w = cs.witness(); dw = ctx.to_device(w)
    r, s = mont(opy, 5), mont(opy, 6)
    ref = key.prove(dw, r, s)
    _, want = orc.groth16_synthetic_expected(7, n_c, 9, 3, r, s)
    assert (ref["Ar"] == want["Ar"]).all() and (ref["Bs"] == want["Bs"]).all() and (ref["Krs"] == want["Krs"]).all()
    for world in (1, 2, 3, 8):
        parts = np.stack([key.prove_shard(dw, rank, world) for rank in range(world)])
        got = key.prove_combine(parts, r, s)
        for k in ("Ar", "Bs", "Krs"):
            assert (got[k] == ref[k]).all(), (world, k)
    # the helper the multi-process path uses, world 1
    from davinci_node_amd import sharding
    got = sharding.sharded_groth16_prove(key, dw, r, s, 0, 1)
    assert all((got[k] == ref[k]).all() for k in ("Ar", "Bs", "Krs"))
    # (keys with small, plain sets: rank 0 computes those MSMs whole -- tests/test_multi_device.py)
    ctx.free(dw); key.free(); cs.free()


# ---------------------------------------------------------------------------------------------------------------------
# One BSB22 commitment: the statetransition shape (8 public inputs + 1 commitment wire).
# What the reference tree pins is restated in oracle/bn254_py.py (challenge formula, where the commitment enters the
# verifier's accumulator, the PoK relation): config/statetransition_vkey.sol:393-416,659-700.
# ---------------------------------------------------------------------------------------------------------------------
def test_keccak256_known_answers(pkg, opy):
    import ctypes as C
    import hashlib
    L = pkg.load_library()
    def lib(b):
        out = C.create_string_buffer(32)
        assert L.dvn_keccak256(b, len(b), out) == 0
        return out.raw
    kat = {b"": "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470",
           b"abc": "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"}
    for msg, h in kat.items():
        assert opy.keccak256(msg).hex() == h and lib(msg).hex() == h
    for n in (1, 55, 96, 135, 136, 137, 271, 272, 273, 1000):            # across block boundaries: library == oracle
        m = bytes((7 * i + n) & 0xFF for i in range(n))
        assert lib(m) == opy.keccak256(m)
    # the oracle's permutation is the SHA-3 one: with SHA-3's domain bits (pad 0x06) it reproduces hashlib.sha3_256
    def sha3_with_oracle_permutation(data):
        rate = 136
        p = bytearray(data) + b"\x06" + bytes((-len(data) - 2) % rate) + b"\x80" if (len(data) + 1) % rate else bytearray(data) + b"\x86"
        st = [[0] * 5 for _ in range(5)]
        for off in range(0, len(p), rate):
            for i in range(rate // 8):
                st[i % 5][i // 5] ^= int.from_bytes(p[off + 8 * i:off + 8 * i + 8], "little")
            st = opy._keccak_f(st)
        return b"".join(st[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    for n in (0, 3, 135, 136, 500):
        assert sha3_with_oracle_permutation(b"q" * n) == hashlib.sha3_256(b"q" * n).digest()


def test_committed_r1cs_matches_oracle_and_verifier_equation(pkg, opy):
    """Host only: the committed synthetic instance equals the oracle's, and the oracle's closed-form proof satisfies the
    verifier's equation A*B = alpha*beta + gamma*(vk_x + C) + delta*Krs in the exponent (statetransition_vkey.sol:393-416:
    the commitment is ADDED to the public-input accumulator that is paired with gamma)."""
    o = opy
    cs = pkg.R1CS(7, 50, 9, n_committed=5)
    ref = o.SyntheticR1CSCommitted(50, 9, 5, 7)
    assert (cs.n_wires, cs.n_public, cs.n_constraints) == (ref.n_wires, 10, 50)
    assert [o.from_mont(o.from_limbs(r), R) for r in cs.witness()] == ref.witness
    cs.free()
    e = o.groth16_committed_expected(ref, 3, 1234, 5678)
    x = e["exponents"]
    assert x["ar"] * x["bs"] % R == (x["alpha"] * x["beta"] + x["gamma"] * x["vkx"] + x["delta"] * x["krs"]) % R
    w = e["witness"]
    assert w[ref.commit_wire] == e["challenge"] != 0
    for A, B, Cc in zip(ref.A, ref.B, ref.C):
        assert sum(c * w[i] for i, c in A) * sum(c * w[i] for i, c in B) % R == sum(c * w[i] for i, c in Cc) % R
    assert e["CommitmentPok"] == o.g1_mul(e["Commitment"], x["sigma"])           # e(C, -sigma G) e(pok, G) = 1


@pytest.mark.gpu
@pytest.mark.parametrize("n_constraints,n_public,n_committed", [(50, 9, 5), (1000, 9, 64), (1 << 12, 9, 700), ((1 << 14) - 3, 4, 1), (1 << 16, 9, 1 << 12)])
def test_groth16_with_commitment_trapdoor_identity(ctx, pkg, opy, n_constraints, n_public, n_committed):
    o = opy
    cs = pkg.R1CS(7, n_constraints, n_public, n_committed=n_committed)
    key = pkg.Groth16Key(ctx, cs, 3)
    ref = o.SyntheticR1CSCommitted(n_constraints, n_public, n_committed, 7)
    assert key.sizes["K"] == ref.n_wires - (n_public + 1) - n_committed
    for r, s in ((1234567, 7654321), (0, 0)):
        want = o.groth16_committed_expected(ref, 3, r, s)
        got = key.prove_committed(cs.witness(), mont(o, r), mont(o, s))
        assert list(got["Commitment"]) == o.g1_to_limbs(want["Commitment"])
        assert list(got["CommitmentPok"]) == o.g1_to_limbs(want["CommitmentPok"])
        assert list(got["challenge"]) == o.limbs(o.to_mont(want["challenge"], R))
        assert list(got["Ar"]) == o.g1_to_limbs(want["Ar"])
        assert list(got["Bs"]) == o.g2_to_limbs(want["Bs"])
        assert list(got["Krs"]) == o.g1_to_limbs(want["Krs"])
    # device-resident witness, and the calldata image of solidity/solidity.go:85-116 with the commitment filled in
    d = ctx.to_device(cs.witness())
    got2 = key.prove_committed(d, mont(o, 1234567), mont(o, 7654321))
    want = o.groth16_committed_expected(ref, 3, 1234567, 7654321)
    assert list(got2["Krs"]) == o.g1_to_limbs(want["Krs"])
    out = np.zeros(384, dtype=np.uint8)
    proof = np.concatenate([got2["Ar"], got2["Bs"], got2["Krs"]])
    assert pkg.load_library().dvn_groth16_proof_abi_encode(proof.ctypes.data, got2["Commitment"].ctypes.data, got2["CommitmentPok"].ctypes.data, out.ctypes.data) == 0
    words = [int.from_bytes(bytes(out[32 * i:32 * i + 32]), "big") for i in range(12)]
    assert words[8:10] == list(want["Commitment"]) and words[10:12] == list(want["CommitmentPok"])
    # a key without commitment refuses the committed entry point, and vice versa the plain prover ignores nothing silently
    ctx.free(d); key.free(); cs.free()
    cs0 = pkg.R1CS(7, 100, 9); key0 = pkg.Groth16Key(ctx, cs0, 3)
    with pytest.raises(pkg.DvnError):
        key0.prove_committed(cs0.witness(), mont(o, 1), mont(o, 2))
    key0.free(); cs0.free()


# ---------------------------------------------------------------------------------------------------------------------
# gnark's own key handed over as arrays + proofs from the solver's output (the drop-in path for the keys the node
# loads at circuits/artifacts.go:391-406): export -> import round trip must reproduce the proof bit for bit.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_constraints", [(1 << 12) - 7, 1 << 16])
def test_imported_key_proves_from_solver_output(ctx, pkg, opy, orc, n_constraints):
    o = opy
    cs = pkg.R1CS(7, n_constraints, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    arrays = key.export_arrays(cs.n_wires, cs.n_public)
    assert arrays["A"].shape[0] == key.sizes["A"] and arrays["B2"].shape == (key.sizes["B"], 16) and arrays["Z"].shape[0] == 1 << key.log_n
    ik = pkg.ImportedKey(ctx, key.log_n, cs.n_wires, arrays)
    ref = o.SyntheticR1CS(n_constraints, 9, 7)
    a, b, c = (np.array([o.limbs(o.to_mont(v, R)) for v in col], dtype=np.uint64) for col in ref.abc())   # the solver's output
    r, s = mont(o, 1234567), mont(o, 7654321)
    got = ik.prove_solved(cs.witness(), a, b, c, r, s)
    want = key.prove(cs.witness(), r, s)
    _, closed = orc.groth16_synthetic_expected(7, n_constraints, 9, 3, r, s)
    for k in ("Ar", "Bs", "Krs"):
        assert (got[k] == want[k]).all() and (got[k] == closed[k]).all(), k
    # the same proof from the shares of worlds 1, 2, 5 (one process per GPU: dvn_groth16_prove_solved_shard + exchange + combine)
    for world in (1, 2, 5):
        parts = np.stack([ik.prove_solved_shard(cs.witness(), a, b, c, rk, world) for rk in range(world)])
        shared = ik.prove_solved_combine(parts, r, s)
        for k in ("Ar", "Bs", "Krs"):
            assert (shared[k] == want[k]).all(), (world, k)
    with pytest.raises(pkg.DvnError):
        ik.prove_solved_shard(cs.witness(), a, b, c, 2, 2)
    # an imported key has no constraint system: the R1CS entry points refuse it; oversized solver output is rejected
    out = np.zeros(32, dtype=np.uint64)
    w = cs.witness()
    rc = ctx.L.dvn_groth16_prove(ctx.h, ik.h, w.ctypes.data, 0, r.ctypes.data, s.ctypes.data, out.ctypes.data)
    assert rc == -1
    big = np.zeros(((1 << key.log_n) + 1, 4), dtype=np.uint64)
    with pytest.raises(pkg.DvnError):
        ik.prove_solved(w, big, big, big, r, s)
    bad = dict(arrays); bad["idxA"] = arrays["idxA"].copy(); bad["idxA"][0] = cs.n_wires
    with pytest.raises(pkg.DvnError):
        pkg.ImportedKey(ctx, key.log_n, cs.n_wires, bad)
    ik.free(); key.free(); cs.free()
