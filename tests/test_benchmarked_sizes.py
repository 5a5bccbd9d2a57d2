"""Every size bench.py quotes a rate for gets a correctness check at THAT size (not only at the small sizes of the other
test files): the proofs and MSMs below run the paths large inputs take -- expanded point tables, four MSM lanes, the staged
sort with its heavy / giant bucket handling -- which the small cases never reach.

The reference's own strongest test is prove -> verify at the circuit's real size
(/root/reference/circuits/test/statetransition/statetransition_test.go:89-250); here the checks are the trapdoor closed
form (no MSM, no NTT in the checker), the C oracle, and the restated Solidity verifier.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
N20 = 1 << 20


def mont(o, x):
    return np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)


@pytest.mark.parametrize("curve", ["bls12377", "bls12381", "bw6761"])
def test_groth16_2p20_trapdoor_identity_other_curves(ctx, pkg, curve):
    """bench extras groth16_synthetic_<curve>_2^20: the voteverifier (BLS12-377) and aggregator (BW6-761) shaped proofs, and
    BLS12-381, at 2^20 constraints against the big-int closed form."""
    from oracle import curves_py as cp
    S = {"bls12377": cp.BLS12_377, "bls12381": cp.BLS12_381, "bw6761": cp.BW6_761}[curve]
    cs = pkg.R1CS(3, N20, 9, curve=curve)
    ocs = cp.SyntheticR1CSOn(S, N20, 9, 3)
    w = cs.witness()
    assert cs.n_wires == ocs.n_wires
    for i in (0, 1, 8, 9, 13, 1000, N20 // 2, cs.n_wires - 1):
        assert S.fr_from_limbs(w[i]) == ocs.witness[i]
    key = pkg.Groth16Key(ctx, cs, 9)
    assert min(key.sizes[k] for k in ("A", "B", "K", "Z", "B2")) >= 1 << 15          # every set is an expanded table
    r, s = S.fr_at(1, 0), S.fr_at(1, 1)
    proof = key.prove(w, np.array(S.fr_to_limbs(r), dtype=np.uint64), np.array(S.fr_to_limbs(s), dtype=np.uint64))
    want = cp.groth16_expected_from_trapdoor_on(ocs, 9, r, s)
    assert list(proof["Ar"]) == S.g1_to_limbs(want["Ar"])
    assert list(proof["Bs"]) == S.g2_to_limbs(want["Bs"])
    assert list(proof["Krs"]) == S.g1_to_limbs(want["Krs"])
    key.free(); cs.free()


def test_committed_statetransition_shape_2p20_closed_form_and_solidity_verifier(ctx, pkg, opy):
    """bench extras groth16_statetransition_shape_2^20: 2^20 constraints, 8 public inputs + ONE BSB22 commitment over 2^14
    committed wires.  Commitment, PoK, challenge and proof equal the closed form; the calldata passes the restated on-chain
    verifier (config/statetransition_vkey.sol)."""
    from oracle import bn254_pairing_py as pp
    o = opy
    n_committed = 1 << 14
    cs = pkg.R1CS(7, N20, 9, n_committed=n_committed)
    key = pkg.Groth16Key(ctx, cs, 3)
    ref = o.SyntheticR1CSCommitted(N20, 9, n_committed, 7)
    r, s = 0xABCDEF123456789, 0x1122334455
    want = o.groth16_committed_expected(ref, 3, r, s)
    got = key.prove_committed(cs.witness(), mont(o, r), mont(o, s))
    assert list(got["Commitment"]) == o.g1_to_limbs(want["Commitment"])
    assert list(got["CommitmentPok"]) == o.g1_to_limbs(want["CommitmentPok"])
    assert list(got["challenge"]) == o.limbs(o.to_mont(want["challenge"], R))
    assert list(got["Ar"]) == o.g1_to_limbs(want["Ar"])
    assert list(got["Bs"]) == o.g2_to_limbs(want["Bs"])
    assert list(got["Krs"]) == o.g1_to_limbs(want["Krs"])
    # Setup -> vk export -> Prove -> ABI calldata -> verifyProof at this size
    vkd = key.verifying_key(cs.n_public, committed=True)
    p1 = lambda l: o.g1_from_limbs([int(v) for v in l])
    p2 = lambda l: o.g2_from_limbs([int(v) for v in l])
    vk = {"ALPHA": p1(vkd["alpha"]), "BETA_NEG": o.g2_neg(p2(vkd["beta"])), "GAMMA_NEG": o.g2_neg(p2(vkd["gamma"])),
          "DELTA_NEG": o.g2_neg(p2(vkd["delta"])), "CONSTANT": p1(vkd["ic"][0]), "PUB": [p1(x) for x in vkd["ic"][1:]],
          "PEDERSEN_G": p2(vkd["pedersen_g"]), "PEDERSEN_GSIGMANEG": p2(vkd["pedersen_gsigmaneg"]), "PUBLIC_COMMITTED": [2]}
    assert vk == want["vk"]
    out = np.zeros(384, dtype=np.uint8)
    proof = np.concatenate([got["Ar"], got["Bs"], got["Krs"]])
    assert pkg.load_library().dvn_groth16_proof_abi_encode(proof.ctypes.data, got["Commitment"].ctypes.data, got["CommitmentPok"].ctypes.data, out.ctypes.data) == 0
    words = [int.from_bytes(bytes(out[32 * i:32 * i + 32]), "big") for i in range(12)]
    inputs = [o.from_mont(o.from_limbs(rw), R) for rw in cs.witness()[1:9]]
    assert pp.verify_proof_sol(vk, words[0:8], words[8:10], words[10:12], inputs, o, o.keccak256)
    bad = list(inputs); bad[0] = (bad[0] + 1) % R
    assert not pp.verify_proof_sol(vk, words[0:8], words[8:10], words[10:12], bad, o, o.keccak256)
    key.free(); cs.free()


def test_imported_key_prove_solved_2p20(ctx, pkg, opy, orc):
    """bench extras groth16_from_host_solution_2^20: the gnark-facing path (key arrays imported, proof from the solver's
    (w, a, b, c) in host memory) at 2^20 against the C oracle's trapdoor closed form."""
    o = opy
    cs = pkg.R1CS(7, N20, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    ik = pkg.ImportedKey(ctx, key.log_n, cs.n_wires, key.export_arrays(cs.n_wires, cs.n_public))
    key.free()
    w = cs.witness()
    # the solver's output for the synthetic system (w[l1] + k w[l2]) * w[r] = w[o]: c is the output wire of each row, b the
    # right factor, a = c / b recovered row by row from the seeded instance
    ref = o.SyntheticR1CS(N20, 9, 7)
    a, b, c = (np.array([o.limbs(o.to_mont(v, R)) for v in col], dtype=np.uint64) for col in ref.abc())
    r, s = mont(o, 1234567), mont(o, 7654321)
    got = ik.prove_solved(w, a, b, c, r, s)
    _, closed = orc.groth16_synthetic_expected(7, N20, 9, 3, r, s)
    for k in ("Ar", "Bs", "Krs"):
        assert (got[k] == closed[k]).all(), k
    ik.free(); cs.free()


def test_msm_g2_2p21_vs_oracle(ctx, orc):
    """A 2^22-constraint proof runs a 2^21-point G2 MSM (bench extras groth16_synthetic_2^22): that size against the C oracle,
    uniform and witness-like scalars."""
    n = 1 << 21
    ph = orc.g2_synthetic_points(5, n)
    pts = ctx.upload_g2(ph)
    for kind in (0, 1):
        sc = orc.scalars(9, kind, n)
        assert (orc.g2_jac_to_affine(ctx.msm_g2(pts, sc)) == orc.msm_g2(ph, sc)).all(), kind
    pts.free()


def test_plonk_shaped_composition_results(ctx, pkg):
    """bench extras plonk_shaped_composition_2^22 (a SYNTHETIC composition: the reference has no PLONK): each of the nine
    2^22 MSMs, three in flight, against the closed form over the arithmetic-progression SRS, and the NTT legs (4 iNTT 2^22,
    8 coset NTT + 1 coset iNTT 2^24) undone by their exact inverses back to the input vector."""
    n = 1 << 22
    pts = ctx.synthetic_g1(0xDA71C1, n)
    vecs = [ctx.synthetic_scalars(100 + k, k % 2, n) for k in range(3)]       # three different scalar vectors, reused 3x
    # closed form per vector: sum s_i (P0 + i D) = (S0 - S1) P0 + S1 P1
    two = ctx.upload_g1(pts.download(0, 2))
    rinv = pow(1 << 256, -1, R)
    to_m = lambda x: [((x << 256) % R >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    want = []
    for d in vecs:
        sm = np.empty((n, 4), dtype=np.uint64); ctx.d2h(sm, d)
        so = sm.astype(object)
        vals = so[:, 0] + (so[:, 1] << 64) + (so[:, 2] << 128) + (so[:, 3] << 192)
        S0 = int(vals.sum()) * rinv % R
        S1 = int((vals * np.arange(n, dtype=object)).sum()) * rinv % R
        want.append(ctx.msm_g1(two, np.array([to_m((S0 - S1) % R), to_m(S1)], dtype=np.uint64)))
        del sm, so, vals
    tk, got = [], []
    for k in range(9):
        tk.append(ctx.msm_g1_submit(pts, vecs[k % 3], n=n))
        if len(tk) == 3:
            got.append(ctx.msm_g1_collect(tk.pop(0)))
    while tk:
        got.append(ctx.msm_g1_collect(tk.pop(0)))
    for k in range(9):
        assert (got[k] == want[k % 3]).all(), k
    two.free(); pts.free()
    for d in vecs[1:]:
        ctx.free(d)
    # NTT legs on a 2^24 vector: forward sequence, then the exact inverses in reverse order
    dom, dom22 = ctx.domain(24), ctx.domain(22)
    dx = ctx.synthetic_scalars(2, 0, 1 << 24)
    x0 = np.empty((1 << 24, 4), dtype=np.uint64); ctx.d2h(x0, dx)
    for _ in range(4):
        ctx.ntt(dom22, dx, True, pkg.DIF)                 # FFTInverse(DIF): natural -> bit-reversed
    for _ in range(8):
        ctx.ntt(dom, dx, False, pkg.DIT, True)            # FFT(DIT, coset): bit-reversed -> natural
    ctx.ntt(dom, dx, True, pkg.DIF, True)                 # FFTInverse(DIF, coset)
    ctx.ntt(dom, dx, False, pkg.DIT, True)                # ... undone
    for _ in range(8):
        ctx.ntt(dom, dx, True, pkg.DIF, True)
    for _ in range(4):
        ctx.ntt(dom22, dx, False, pkg.DIT)
    x1 = np.empty_like(x0); ctx.d2h(x1, dx)
    assert (x0 == x1).all()
    ctx.free(dx); ctx.free(vecs[0]); dom.free(); dom22.free()


# ---------------------------------------------------------------------------------------------------------------------
# Above 2^22.  The statetransition circuit's real constraint count is not known offline (its artifacts are CDN-hosted), so the
# path is parametric in log2 n up to 24: these run the same checks at the top of that range.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [22, 24])
def test_compute_h_2p22_2p24_vs_oracle(ctx, orc, log_n):
    """computeH (7 transforms + the pointwise pass, a / b / c resident) at 2^22 and 2^24 against the C oracle, bit for bit, with
    ragged input lengths (the domain zero-pads)."""
    m = (1 << log_n) - 5
    dom = ctx.domain(log_n)
    a, b, c = orc.scalars(3, 0, m), orc.scalars(4, 0, m), orc.scalars(5, 0, m)
    got = ctx.compute_h(dom, a, b, c)
    want = orc.compute_h(a, b, c, log_n, orc.usable_cores())
    assert (got == want).all()
    dom.free()


def _closed_form(ctx, pts, dscal, n):
    """sum s_i (P0 + i D) = (S0 - S1) P0 + S1 P1 over the arithmetic-progression point set, as a 2-point MSM."""
    sm = np.empty((n, 4), dtype=np.uint64); ctx.d2h(sm, dscal)
    idx = np.arange(n, dtype=np.uint64)
    s0 = s1 = 0
    for limb in range(4):
        for piece in range(4):                                   # 16-bit digit planes keep every partial sum below 2^64
            d = (sm[:, limb] >> np.uint64(16 * piece)) & np.uint64(0xFFFF)
            sh = 64 * limb + 16 * piece
            s0 += int(d.sum(dtype=np.uint64)) << sh
            s1 += int(np.dot(idx, d)) << sh
    rinv = pow(1 << 256, -1, R)
    S0, S1 = s0 * rinv % R, s1 * rinv % R
    to_m = lambda x: [((x << 256) % R >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    two = ctx.upload_g1(pts.download(0, 2))
    out = ctx.msm_g1(two, np.array([to_m((S0 - S1) % R), to_m(S1)], dtype=np.uint64))
    two.free()
    return out


def test_msm_g1_2p24_closed_form(ctx, pkg):
    """One G1 MSM over 2^24 points (a 14 GB table of 13 rows; 2^27.7 sorted entries) against the closed form, uniform and
    witness-like scalars, and three in flight."""
    n = 1 << 24
    pts = ctx.synthetic_g1(0xDA71C1, n)
    c, nwin, combine = ctx.msm_plan_points(pts, n)
    assert combine == 0 and c == 20 and nwin == 13
    for kind in (0, 1):
        d = ctx.synthetic_scalars(7 + kind, kind, n)
        want = _closed_form(ctx, pts, d, n)
        assert (ctx.msm_g1(pts, d, n=n) == want).all(), kind
        if kind == 0:
            tk = [ctx.msm_g1_submit(pts, d, n=n) for _ in range(3)]
            assert all((ctx.msm_g1_collect(t) == want).all() for t in tk)
        ctx.free(d)
    pts.free()


@pytest.mark.parametrize("log_n", [23, 24])
def test_groth16_2p23_2p24_trapdoor_identity(ctx, pkg, opy, orc, log_n):
    """The synthetic statetransition-shaped proof at 2^23 and 2^24 constraints (key sets of that many points as 13-row tables,
    computeH on that domain) against the trapdoor closed form."""
    n_c = 1 << log_n
    cs = pkg.R1CS(7, n_c, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    assert key.log_n == log_n
    r, s = mont(opy, 5), mont(opy, 7)
    dw = ctx.to_device(cs.witness())
    got = key.prove(dw, r, s)
    _, want = orc.groth16_synthetic_expected(7, n_c, 9, 3, r, s)
    for k in ("Ar", "Bs", "Krs"):
        assert (got[k] == want[k]).all(), k
    ctx.free(dw); key.free(); cs.free()


def test_sizes_beyond_31_bit_entry_positions_are_refused(ctx, pkg):
    """Entry positions and point references are 31-bit: a set whose table would need more rows x points than that stays plain,
    an MSM whose windows x n does not fit is refused with DVN_ERR_INVALID_ARG (a message that names the limit), and the
    context keeps working."""
    n = 1 << 28                                                  # 16 GiB of points: fits HBM, 16 windows x 2^28 entries do not fit 31 bits
    pts = ctx.synthetic_g1(3, n)
    c, nwin, combine = ctx.msm_plan_points(pts, n)
    assert combine != 0                                           # left plain: 13 rows x 2^28 references would overflow
    d = ctx.synthetic_scalars(1, 0, 1 << 20)
    big = ctx.synthetic_scalars(2, 1, n)
    with pytest.raises(pkg.DvnError) as e:
        ctx.msm_g1_submit(pts, big, n=n)
    assert e.value.code == -1 and "31-bit" in str(e.value)
    ctx.free(big)
    with pytest.raises(pkg.DvnError) as e:
        pts.set_expanded(20)
    assert e.value.code == -1 and "too large" in str(e.value)
    # a 2^20 slice of the same set still runs (plain path), and equals the closed form
    assert (ctx.msm_g1(pts, d, n=1 << 20) == _closed_form(ctx, pts, d, 1 << 20)).all()
    ctx.free(d); pts.free()
