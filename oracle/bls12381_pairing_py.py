"""TEST INFRASTRUCTURE ONLY -- BLS12-381 ate pairing in Python big ints, to VERIFY the blob-KZG proofs of the HIP path against the
G2 points of the trusted setup the reference embeds (/root/reference/config/kzg_trusted_setup.txt holds [tau^j]G2, j <= 64):

  opening proof   e(pi, [tau]_2 - [z]_2) == e(C - [y]_1, [1]_2)                     (EIP-4844 verify_kzg_proof_impl; the same equation the
                                                                                 reference's in-circuit verifier checks, crypto/blobs/kzg.go)
  cell proof      e(pi_k, [tau^64]_2 - [h_k^64]_2) == e(C - [I_k(tau)]_1, [1]_2)    (EIP-7594 verify_cell_kzg_proof_batch for one cell)

so proofs are accepted by the ceremony's own G2 side -- independent of the oracle's quotient arithmetic.  Construction: Fp12 as
Fp[w]/(w^12 - 2 w^6 + 2) (w^6 = u + 1), M-type twist untwisted by (x / w^2, y / w^3), Miller loop over |x| = 0xd201000000010000
without sign handling (a fixed power of the pairing: still bilinear and non-degenerate, which is all a product check needs),
final exponentiation (p^12 - 1)/r.  Pinned by bilinearity / order / non-degeneracy (tests/test_kzg.py).  Only tests/ may import it.
"""
from __future__ import annotations

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ATE_LOOP = 15132376222941642752
LOG_ATE = 62
MODULUS_COEFFS = [2, 0, 0, 0, 0, 0, -2, 0, 0, 0, 0, 0]        # w^12 = 2 w^6 - 2


class F12:
    """Fp[w]/(w^12 - 2 w^6 + 2), coefficients little-endian."""
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
        for k in range(22, 11, -1):                 # w^k = w^(k-12) * (2 w^6 - 2)
            v = t[k]
            if v:
                t[k - 6] += 2 * v
                t[k - 12] -= 2 * v
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
_W2I, _W3I = (W * W).inv(), (W * W * W).inv()


def cast_g1(pt):
    return (F12([pt[0]] + [0] * 11), F12([pt[1]] + [0] * 11))


def twist(q):
    """G2 point over Fp2 = Fp[u]/(u^2 + 1), (a0, a1) = a0 + a1 u, into E(Fp12): u -> w^6 - 1, then (x / w^2, y / w^3)."""
    (x0, x1), (y0, y1) = q
    nx = F12([x0 - x1] + [0] * 5 + [x1] + [0] * 5)
    ny = F12([y0 - y1] + [0] * 5 + [y1] + [0] * 5)
    return (nx * _W2I, ny * _W3I)


def _double(pt):
    x, y = pt
    m = (x * x).scale(3) / y.scale(2)
    nx = m * m - x.scale(2)
    return (nx, m * (x - nx) - y)


def _add(p1, p2):
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _line(p1, p2, t):
    x1, y1 = p1; x2, y2 = p2; xt, yt = t
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1).scale(3) / y1.scale(2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(q2, p1):
    """q2: G2 affine ((x0, x1), (y0, y1)) or None; p1: G1 affine (x, y) or None."""
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
    return f


def final_exp(f):
    return f ** ((P ** 12 - 1) // R)


def pairing(q2, p1):
    return final_exp(miller_loop(q2, p1))


def pairing_check(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 for pairs (G1 point, G2 point); None = infinity."""
    f = F12.one()
    for p1, q2 in pairs:
        f = f * miller_loop(q2, p1)
    return final_exp(f) == F12.one()


# ---- G2 of the setup file: 96-byte ZCash compression (x.c1 || x.c0, flags in the first byte) ------------------------------
def _fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _fp2_sqrt(a):
    """Square root in Fp[u]/(u^2 + 1), p = 3 mod 4 (complex method); None if a is not a square."""
    if a == (0, 0):
        return (0, 0)
    a0, a1 = a
    if a1 == 0:
        r = pow(a0, (P + 1) // 4, P)
        if r * r % P == a0:
            return (r, 0)
        r = pow(-a0 % P, (P + 1) // 4, P)
        return (0, r) if r * r % P == -a0 % P else None
    norm = (a0 * a0 + a1 * a1) % P
    n = pow(norm, (P + 1) // 4, P)
    if n * n % P != norm:
        return None
    for sgn in (1, -1):
        t = (a0 + sgn * n) * pow(2, -1, P) % P
        x0 = pow(t, (P + 1) // 4, P)
        if x0 * x0 % P == t and x0:
            x1 = a1 * pow(2 * x0, -1, P) % P
            if _fp2_mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


def g2_decompress(b96: bytes):
    assert len(b96) == 96 and b96[0] & 0x80
    if b96[0] & 0x40:
        return None
    x1 = int.from_bytes(bytes([b96[0] & 0x1F]) + b96[1:48], "big")
    x0 = int.from_bytes(b96[48:], "big")
    x = (x0, x1)
    y2 = _fp2_mul(_fp2_mul(x, x), x)
    y2 = ((y2[0] + 4) % P, (y2[1] + 4) % P)                 # E': y^2 = x^3 + 4 (u + 1)
    y = _fp2_sqrt(y2)
    assert y is not None, "x is not on the twist"
    largest = y[1] > (P - 1) // 2 if y[1] else y[0] > (P - 1) // 2
    if largest != bool(b96[0] & 0x20):
        y = ((-y[0]) % P, (-y[1]) % P)
    return (x, y)


def g2_add(a, b):
    """Affine addition on the twist over Fp2 (None = infinity)."""
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    inv = lambda v: (lambda d: ((v[0] * d) % P, (-v[1] * d) % P))(pow((v[0] * v[0] + v[1] * v[1]) % P, -1, P))
    sub = lambda u, v: ((u[0] - v[0]) % P, (u[1] - v[1]) % P)
    if x1 == x2:
        if y1 != y2 or y1 == (0, 0):
            return None
        m = _fp2_mul(_fp2_mul((3, 0), _fp2_mul(x1, x1)), inv(_fp2_mul((2, 0), y1)))
    else:
        m = _fp2_mul(sub(y2, y1), inv(sub(x2, x1)))
    x3 = sub(sub(_fp2_mul(m, m), x1), x2)
    return (x3, sub(_fp2_mul(m, sub(x1, x3)), y1))


def g2_neg(a):
    return None if a is None else (a[0], ((-a[1][0]) % P, (-a[1][1]) % P))


def g2_mul(a, k: int):
    acc, base = None, a
    k %= R
    while k:
        if k & 1:
            acc = g2_add(acc, base)
        base = g2_add(base, base)
        k >>= 1
    return acc
