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
    # a key with small (plain) sets cannot be row-sharded
    cs2 = pkg.R1CS(7, 1000, 3); key2 = pkg.Groth16Key(ctx, cs2, 3)
    with pytest.raises(pkg.DvnError):
        key2.prove_shard(cs2.witness(), 0, 2)
    key2.free(); cs2.free(); ctx.free(dw); key.free(); cs.free()
