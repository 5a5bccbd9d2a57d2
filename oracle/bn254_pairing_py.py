"""TEST INFRASTRUCTURE ONLY -- BN254 optimal-ate pairing and the reference's Groth16 verifier, in Python big ints.

Why: every prover test of the reference is a prove -> verify round trip, the strongest one through the
generated Solidity verifier on a simulated EVM (circuits/test/statetransition/statetransition_test.go:198-249).  That
verifier is IN THE TREE: /root/reference/config/statetransition_vkey.sol.  `verify_proof_sol` below restates its
`verifyProof` (lines 654-744) + `publicInputMSM` (393-520) step by step over the EVM precompile semantics
(ECADD / ECMUL / ECPAIRING = EIP-196 / EIP-197), taking the verifying key as data instead of as Solidity constants, so the
HIP prover's output -- proof, commitment, proof of knowledge, in their calldata encoding -- is checked by the reference's own
verification algorithm, not only by the trapdoor identity.

The pairing itself (EIP-197's, un-vendored in the reference: the EVM precompile) is the textbook construction: Fp12 as
Fp[w]/(w^12 - 18 w^6 + 82), the sextic twist embedding, a Miller loop over 6x+2 with the two Frobenius line corrections
and the final exponentiation (p^12 - 1)/r.  It is pinned here by bilinearity and non-degeneracy on the reference's own
verifying-key points (tests/test_pairing.py); any non-degenerate bilinear map on G1 x G2 decides the verifier's product
equation identically.  Only tests/ may import this module.
"""
from __future__ import annotations

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ATE_LOOP = 29793968203157093288            # 6x + 2, x = 4965661367192848881
LOG_ATE = 63
MODULUS_COEFFS = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]      # w^12 = 18 w^6 - 82


class F12:
    """Fp[w]/(w^12 - 18 w^6 + 82), coefficients little-endian."""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [x % P for x in c]

    @staticmethod
    def one():
        return F12([1] + [0] * 11)

    @staticmethod
    def zero():
        return F12([0] * 12)

    def __add__(self, o):
        return F12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return F12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return F12([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def scale(self, k):
        return F12([a * k for a in self.c])

    def __mul__(self, o):
        if isinstance(o, int):
            return self.scale(o)
        t = [0] * 23
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(o.c):
                    t[i + j] += a * b
        for k in range(22, 11, -1):                 # w^k = w^(k-12) * (18 w^6 - 82)
            v = t[k]
            if v:
                t[k - 6] += 18 * v
                t[k - 12] -= 82 * v
        return F12(t[:12])

    def is_zero(self):
        return not any(self.c)

    def inv(self):
        """Extended Euclid on polynomials over Fp."""
        def deg(p):
            d = len(p) - 1
            while d and p[d] % P == 0:
                d -= 1
            return d

        def poly_div(a, b):                          # quotient of a / b (rounded), py_ecc's poly_rounded_div
            dega, degb = deg(a), deg(b)
            temp = list(a); o = [0] * len(a)
            for i in range(dega - degb, -1, -1):
                q = temp[degb + i] * pow(b[degb], -1, P) % P
                o[i] = (o[i] + q) % P
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - b[c] * q) % P
            return [x % P for x in o[:deg(o) + 1]]

        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], [x % P for x in MODULUS_COEFFS] + [1]
        while deg(low):
            r = poly_div(high, low)
            r += [0] * (13 - len(r))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                    new[i + j] = (new[i + j] - low[i] * r[j]) % P
            lm, low, hm, high = nm, new, lm, low
        li = pow(low[0], -1, P)
        return F12([x * li for x in lm[:12]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e):
        out, base = F12.one(), self
        while e:
            if e & 1:
                out = out * base
            base = base * base
            e >>= 1
        return out


W = F12([0, 1] + [0] * 10)


def cast_g1(pt):
    return (F12([pt[0]] + [0] * 11), F12([pt[1]] + [0] * 11))


def twist(q):
    """G2 point over Fp2 = Fp[u]/(u^2 + 1), coordinates (a0, a1) = a0 + a1 u, into E(Fp12): u -> w^6 - 9."""
    (x0, x1), (y0, y1) = q
    nx = F12([x0 - 9 * x1] + [0] * 5 + [x1] + [0] * 5)
    ny = F12([y0 - 9 * y1] + [0] * 5 + [y1] + [0] * 5)
    return (nx * (W * W), ny * (W * W * W))


def _double(pt):
    x, y = pt
    m = (x * x).scale(3) / y.scale(2)
    nx = m * m - x.scale(2)
    return (nx, m * (x - nx) - y)


def _add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _line(p1, p2, t):
    """Line through p1, p2 (tangent when equal) evaluated at t."""
    x1, y1 = p1; x2, y2 = p2; xt, yt = t
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1).scale(3) / y1.scale(2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(q2, p1):
    """q2: G2 affine ((x0, x1), (y0, y1)) or None; p1: G1 affine (x, y) or None.  No final exponentiation."""
    if q2 is None or p1 is None:
        return F12.one()
    Q, Pt = twist(q2), cast_g1(p1)
    Rp, f = Q, F12.one()
    for i in range(LOG_ATE, -1, -1):
        f = f * f * _line(Rp, Rp, Pt)
        Rp = _double(Rp)
        if ATE_LOOP & (1 << i):
            f = f * _line(Rp, Q, Pt)
            Rp = _add(Rp, Q)
    Q1 = (Q[0] ** P, Q[1] ** P)
    nQ2 = (Q1[0] ** P, -(Q1[1] ** P))
    f = f * _line(Rp, Q1, Pt)
    Rp = _add(Rp, Q1)
    f = f * _line(Rp, nQ2, Pt)
    return f


def final_exp(f):
    return f ** ((P ** 12 - 1) // R)


def pairing(q2, p1):
    return final_exp(miller_loop(q2, p1))


def pairing_check(pairs) -> bool:
    """EIP-197 ECPAIRING: prod e(P_i, Q_i) == 1 for pairs (G1 point, G2 point); None = infinity."""
    f = F12.one()
    for p1, q2 in pairs:
        f = f * miller_loop(q2, p1)
    return final_exp(f) == F12.one()


# ---- the reference's Solidity verifier, restated -------------------------------------------------------------------------
def verify_proof_sol(vk: dict, proof8, commitments2, pok2, inputs, ec, keccak256) -> bool:
    """config/statetransition_vkey.sol verifyProof(proof, commitments, commitmentPok, input), lines 654-744.

    vk: the constants the .sol file hard-codes -- ALPHA (G1), BETA_NEG / GAMMA_NEG / DELTA_NEG (G2, A0/A1 order), CONSTANT
    and PUB[0..n] (G1; the last one belongs to the commitment wire), PEDERSEN_G / PEDERSEN_GSIGMANEG (G2),
    PUBLIC_COMMITTED (0-based input indices hashed into the challenge: [2] in the file, line 665).
    proof8: the calldata words [A.x, A.y, B.x1, B.x0, B.y1, B.y0, C.x, C.y] (EIP-197 order: A1 before A0).
    ec: module with g1_add / g1_mul / g1_neg over affine int pairs (oracle/bn254_py.py); keccak256: bytes -> bytes.
    """
    rmod = R
    # :659-676 HashToField: publicCommitments[0] = keccak256(commitments || input[committed]) % R
    data = commitments2[0].to_bytes(32, "big") + commitments2[1].to_bytes(32, "big")
    data += b"".join(inputs[i].to_bytes(32, "big") for i in vk["PUBLIC_COMMITTED"])
    public_commitment = int.from_bytes(keccak256(data), "big") % rmod
    cpt = None if commitments2 == [0, 0] else (commitments2[0], commitments2[1])
    kpt = None if pok2 == [0, 0] else (pok2[0], pok2[1])
    # :678-700 Pedersen proof of knowledge: e(commitment, GSIGMANEG) * e(pok, G) == 1
    if not pairing_check([(cpt, vk["PEDERSEN_GSIGMANEG"]), (kpt, vk["PEDERSEN_G"])]):
        return False                                                       # CommitmentInvalid
    # :393-520 publicInputMSM: CONSTANT + commitment + sum input_i * PUB_i + publicCommitment * PUB_last
    if any(s >= rmod for s in inputs):
        return False                                                       # PublicInputNotInField
    acc = ec.g1_add(vk["CONSTANT"], cpt)
    for s, pub in zip(list(inputs) + [public_commitment], vk["PUB"]):
        acc = ec.g1_add(acc, ec.g1_mul(pub, s))
    # :706-744 e(A, B) * e(C, -delta) * e(alpha, -beta) * e(vk_x, -gamma) == 1
    a = (proof8[0], proof8[1]); c = (proof8[6], proof8[7])
    b = ((proof8[3], proof8[2]), (proof8[5], proof8[4]))                   # calldata holds A1 first
    return pairing_check([(a, b), (c, vk["DELTA_NEG"]), (vk["ALPHA"], vk["BETA_NEG"]), (acc, vk["GAMMA_NEG"])])


def verify_groth16(vk: dict, a, b, c, inputs, ec) -> bool:
    """gnark groth16.Verify without commitments (what CircuitRuntime.Verify forwards to, /root/reference/circuits/artifacts.go:595-613):
    e(A, B) == e(alpha, beta) e(vk_x, gamma) e(C, delta), vk_x = IC_0 + sum input_i IC_(i+1); vk holds the negated G2 points."""
    acc = vk["CONSTANT"]
    for s, pub in zip(inputs, vk["PUB"]):
        acc = ec.g1_add(acc, ec.g1_mul(pub, s))
    return pairing_check([(a, b), (c, vk["DELTA_NEG"]), (vk["ALPHA"], vk["BETA_NEG"]), (acc, vk["GAMMA_NEG"])])
