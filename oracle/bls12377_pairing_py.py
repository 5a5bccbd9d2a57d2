"""TEST INFRASTRUCTURE ONLY -- BLS12-377 ate pairing in Python big ints, so that proofs of the voteverifier curve
(/root/reference/spec/params/params.go:21-27; 60 such proofs per batch) get the same prove -> verify round trip the reference's
tests perform.  Fp2 = Fp[u]/(u^2 + 5), Fp12 = Fp[w]/(w^12 + 5) with w^6 = u, D-type twist y^2 = x^3 + 1/u untwisted by
(x w^2, y w^3), Miller loop over x = 0x8508c00000000001, final exponentiation (p^12 - 1)/r.  Pinned by bilinearity, order and
non-degeneracy only (the reference holds no BLS12-377 data): tests/test_groth16_curves.py.  Only tests/ may import it.
"""
from __future__ import annotations

P = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
R = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
ATE_LOOP = 0x8508C00000000001
LOG_ATE = 62
MODULUS_COEFFS = [5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]         # w^12 = -5


class F12:
    """Fp[w]/(w^12 + 5), coefficients little-endian."""
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
        for k in range(22, 11, -1):                 # w^k = -5 w^(k-12)
            v = t[k]
            if v:
                t[k - 12] -= 5 * v
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
_W2, _W3 = W * W, W * W * W


def cast_g1(pt):
    return (F12([pt[0]] + [0] * 11), F12([pt[1]] + [0] * 11))


def twist(q):
    """G2 point over Fp2 = Fp[u]/(u^2 + 5), (a0, a1) = a0 + a1 u, into E(Fp12): u -> w^6, then (x w^2, y w^3)."""
    (x0, x1), (y0, y1) = q
    nx = F12([x0] + [0] * 5 + [x1] + [0] * 5)
    ny = F12([y0] + [0] * 5 + [y1] + [0] * 5)
    return (nx * _W2, ny * _W3)


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
