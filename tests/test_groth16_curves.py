"""Groth16 on the voteverifier curve (BLS12-377) and on BLS12-381 (prover_gpu.go:45-50): the curve-generic Setup / Prove mirrors
(csrc/groth16_impl.h) against the closed form a known trapdoor gives (oracle/curves_py.py).  Parity unpinned by the
reference (no vectors); the trapdoor identity is exact: Ar = [alpha + A(tau) + r delta] G1 etc."""
from math import isqrt

import numpy as np
import pytest


@pytest.fixture(scope="module", params=["bls12377", "bls12381", "bw6761"])
def S(request):
    from oracle import curves_py
    s = {"bls12377": curves_py.BLS12_377, "bls12381": curves_py.BLS12_381, "bw6761": curves_py.BW6_761}[request.param]
    s.cname = request.param
    return s


def test_bls12381_g2_generator_is_r_torsion():
    from oracle import curves_py
    S = curves_py.BLS12_381
    assert S.g2_on_curve(S.g2_gen) and S.g2_mul(S.g2_gen, S.r) is None


def test_bls12377_g2_generator_is_r_torsion():
    """The synthetic setup's twist generator: re-derive [#E'(Fp2)/r] * Q from the curve's trace."""
    from oracle import curves_py
    S = curves_py.BLS12_377
    p, r = S.p, S.r
    x = 0x8508C00000000001                                   # the BLS12-377 seed: t = x + 1
    t = x + 1
    assert (p + 1 - t) % r == 0
    t2 = t * t - 2 * p
    f2 = isqrt((4 * p * p - t2 * t2) // 3)
    assert 3 * f2 * f2 == 4 * p * p - t2 * t2
    n_twist = p * p + 1 - (t2 - 3 * f2) // 2
    assert n_twist % r == 0
    q = S.g2_some_point(1)
    assert S.g2_mul(q, n_twist) is None                      # the order of the twist really is n_twist
    g = S.g2_mul(q, n_twist // r)
    assert g == S.g2_gen and S.g2_on_curve(g) and S.g2_mul(g, r) is None


def test_bw6761_generators_are_r_torsion():
    """Both BW6-761 groups live over Fp: find the CM trace (r | p + 1 - t, 4p - t^2 = 3 f^2), the group orders among the six
    twist traces, and re-derive the two synthetic-setup generators by cofactor clearing."""
    from oracle import curves_py
    W = curves_py.BW6_761
    p, r = W.p, W.r
    lim = 2 * isqrt(p) + 2
    traces = []
    k = -(lim // r) - 2
    while True:
        t = (p + 1) % r + k * r
        if t > lim:
            break
        d = 4 * p - t * t
        if abs(t) <= lim and d > 0 and d % 3 == 0 and 3 * isqrt(d // 3) ** 2 == d:
            traces.append((t, isqrt(d // 3)))
        k += 1
    assert traces
    t, f = traces[0]
    cands = [t, -t, (t + 3 * f) // 2, (t - 3 * f) // 2, (-t + 3 * f) // 2, (-t - 3 * f) // 2]
    for group, gen in ((1, W.g1_gen), (2, W.g2_gen)):
        q = W.some_point(group)
        orders = [p + 1 - tt for tt in cands if W.mul(q, p + 1 - tt) is None]
        assert len(orders) == 1 and orders[0] % r == 0
        assert W.mul(q, orders[0] // r) == gen and W.on_curve(group, gen) and W.mul(gen, r) is None


def test_generic_oracle_matches_bn254_oracle(opy):
    """curves_py's suite-generic trapdoor proof == bn254_py's on BN254 (same streams, same closed form)."""
    from oracle import curves_py as cp
    cp.BN254.g2_gen = opy.G2_GEN
    cs_g = cp.SyntheticR1CSOn(cp.BN254, 20, 3, 7)
    cs_b = opy.SyntheticR1CS(20, 3, 7)
    assert cs_g.witness == cs_b.witness and cs_g.A == cs_b.A
    pk, trap = opy.groth16_setup(cs_b, 9)
    want = opy.groth16_expected_from_trapdoor(cs_b, trap, pk["domain"], 5, 11)
    got = cp.groth16_expected_from_trapdoor_on(cs_g, 9, 5, 11)
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("n_constraints,n_public", [(5, 1), (300, 3), (3000, 2)])
def test_groth16_trapdoor_identity_other_curves(ctx, pkg, S, n_constraints, n_public):
    from oracle import curves_py as cp
    cs = pkg.R1CS(3, n_constraints, n_public, curve=S.cname)
    ocs = cp.SyntheticR1CSOn(S, n_constraints, n_public, 3)
    w = cs.witness()
    assert cs.n_wires == ocs.n_wires
    assert [S.fr_from_limbs(x) for x in w[:50]] == ocs.witness[:50]
    key = pkg.Groth16Key(ctx, cs, 9)
    for r, s in ((5, 11), (S.fr_at(1, 0), S.fr_at(1, 1))):
        proof = key.prove(w, np.array(S.fr_to_limbs(r), dtype=np.uint64), np.array(S.fr_to_limbs(s), dtype=np.uint64))
        want = cp.groth16_expected_from_trapdoor_on(ocs, 9, r, s)
        assert list(proof["Ar"]) == S.g1_to_limbs(want["Ar"])
        assert list(proof["Bs"]) == S.g2_to_limbs(want["Bs"])
        assert list(proof["Krs"]) == S.g1_to_limbs(want["Krs"])
    # a BN254 constraint system is rejected by this curve's setup, and vice versa
    cs_bn = pkg.R1CS(3, 10, 1)
    with pytest.raises(pkg.DvnError):
        ctx._chk(getattr(ctx.L, f"dvn_{S.cname}_groth16_setup")(ctx.h, cs_bn.h, 9, __import__("ctypes").byref(__import__("ctypes").c_void_p())))
    cs_bn.free(); key.free(); cs.free()


@pytest.mark.gpu
def test_imported_key_on_other_curves(ctx, pkg, S):
    """The gnark-key path (pk_import + prove_solved) on the voteverifier / aggregator / BLS12-381 curves: a key
    exported from the Setup mirror and imported again proves from the solver's (w, a, b, c) to the same proof and to the trapdoor
    identity."""
    from oracle import curves_py as cp
    n_constraints, n_public = 700, 3
    cs = pkg.R1CS(3, n_constraints, n_public, curve=S.cname)
    ocs = cp.SyntheticR1CSOn(S, n_constraints, n_public, 3)
    key = pkg.Groth16Key(ctx, cs, 9)
    ik = pkg.ImportedKey(ctx, key.log_n, cs.n_wires, key.export_arrays(cs.n_wires, cs.n_public), curve=S.cname)
    w = cs.witness()
    ev = lambda rows: np.array([S.fr_to_limbs(sum(c * ocs.witness[i] for i, c in row) % S.r) for row in rows], dtype=np.uint64)
    r, s = S.fr_at(1, 0), S.fr_at(1, 1)
    rl, sl = np.array(S.fr_to_limbs(r), dtype=np.uint64), np.array(S.fr_to_limbs(s), dtype=np.uint64)
    got = ik.prove_solved(w, ev(ocs.A), ev(ocs.B), ev(ocs.C), rl, sl)
    want = key.prove(w, rl, sl)
    closed = cp.groth16_expected_from_trapdoor_on(ocs, 9, r, s)
    for k in ("Ar", "Bs", "Krs"):
        assert (got[k] == want[k]).all(), k
    assert list(got["Ar"]) == S.g1_to_limbs(closed["Ar"]) and list(got["Krs"]) == S.g1_to_limbs(closed["Krs"])
    ik.free(); key.free(); cs.free()
    if S.cname == "bls12381":
        return
    # one process per GPU on the voteverifier / aggregator curves: the shares of worlds 1, 2, 5 (dvn_<curve>_groth16_prove_solved_shard,
    # key sets large enough to be shared by entry count) assemble to the same proof (dvn_<curve>_groth16_prove_combine)
    n_constraints = 5000
    cs = pkg.R1CS(4, n_constraints, n_public, curve=S.cname)
    ocs = cp.SyntheticR1CSOn(S, n_constraints, n_public, 4)
    key = pkg.Groth16Key(ctx, cs, 9)
    assert min(key.sizes["A"], key.sizes["B"]) >= 1 << 10
    ik = pkg.ImportedKey(ctx, key.log_n, cs.n_wires, key.export_arrays(cs.n_wires, cs.n_public), curve=S.cname)
    w = cs.witness(); a, b, c = ev(ocs.A), ev(ocs.B), ev(ocs.C)
    whole = ik.prove_solved(w, a, b, c, rl, sl)
    for world in (1, 2, 5):
        parts = np.stack([ik.prove_solved_shard(w, a, b, c, rk, world) for rk in range(world)])
        assert parts.shape[1] == {"bls12377": 108, "bw6761": 180}[S.cname]
        shared = ik.prove_solved_combine(parts, rl, sl)
        for k in ("Ar", "Bs", "Krs"):
            assert (shared[k] == whole[k]).all(), (world, k)
    with pytest.raises(pkg.DvnError):
        ik.prove_solved_shard(w, a, b, c, 5, 5)
    ik.free(); key.free(); cs.free()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12381", "bls12377", "bw6761"])
def test_device_proof_passes_groth16_verify(ctx, pkg, curve):
    """Prove -> verify on BLS12-381 and on the voteverifier curve BLS12-377 with the library's own verifying key:
    e(A, B) == e(alpha, beta) e(vk_x, gamma) e(C, delta) over big-int pairings (the BLS12-381 one is pinned on the reference's
    trusted setup in tests/test_kzg.py, the BLS12-377 one by bilinearity / order / non-degeneracy here)."""
    from oracle import curves_py as cp
    if curve == "bls12381":
        from oracle import bls12381_pairing_py as BP
        S = cp.BLS12_381
    elif curve == "bls12377":
        from oracle import bls12377_pairing_py as BP
        S = cp.BLS12_377
        e = BP.pairing(S.g2_gen, S.g1_gen)
        assert not (e == BP.F12.one()) and e ** S.r == BP.F12.one()
        assert BP.pairing(S.g2_mul(S.g2_gen, 5), S.g1_mul(S.g1_gen, 3)) == e ** 15
    else:                                         # the aggregator curve: embedding degree 6, both groups over Fp
        from oracle import bw6761_pairing_py as BP
        S = cp.BW6_761
        e = BP.pairing(S.g2_gen, S.g1_gen)
        assert not (e == BP.F6.one()) and e ** S.r == BP.F6.one()
        assert BP.pairing(S.g2_mul(S.g2_gen, 5), S.g1_mul(S.g1_gen, 3)) == e ** 15
    n_constraints, n_public = 400, 4
    cs = pkg.R1CS(3, n_constraints, n_public, curve=curve)
    key = pkg.Groth16Key(ctx, cs, 9)
    vk = key.verifying_key(n_public)
    fw = 12 if curve == "bw6761" else 6           # uint64 words per base-field element
    rinv = pow(1 << (64 * fw), -1, S.p)
    def fp(words):
        return int(sum(int(w) << (64 * i) for i, w in enumerate(words))) * rinv % S.p
    def g1(l):
        l = [int(v) for v in l]
        return None if not any(l) else (fp(l[0:fw]), fp(l[fw:2 * fw]))
    def g2(l):
        l = [int(v) for v in l]
        if curve == "bw6761":
            return None if not any(l) else (fp(l[0:fw]), fp(l[fw:2 * fw]))
        return None if not any(l) else ((fp(l[0:6]), fp(l[6:12])), (fp(l[12:18]), fp(l[18:24])))
    w = cs.witness()
    r, s = S.fr_at(2, 0), S.fr_at(2, 1)
    proof = key.prove(w, np.array(S.fr_to_limbs(r), dtype=np.uint64), np.array(S.fr_to_limbs(s), dtype=np.uint64))
    inputs = [S.fr_from_limbs(x) for x in w[1:n_public]]
    acc = g1(vk["ic"][0])
    for v, row in zip(inputs, vk["ic"][1:]):
        acc = S.g1_add(acc, S.g1_mul(g1(row), v))
    neg = lambda p: (p[0], (-p[1]) % S.p)
    a, b, c = g1(proof["Ar"]), g2(proof["Bs"]), g1(proof["Krs"])
    assert BP.pairing_check([(a, b), (neg(c), g2(vk["delta"])), (neg(g1(vk["alpha"])), g2(vk["beta"])), (neg(acc), g2(vk["gamma"]))])
    acc_bad = S.g1_add(acc, S.g1_gen)
    assert not BP.pairing_check([(a, b), (neg(c), g2(vk["delta"])), (neg(g1(vk["alpha"])), g2(vk["beta"])), (neg(acc_bad), g2(vk["gamma"]))])
    key.free(); cs.free()
