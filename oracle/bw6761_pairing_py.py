"""TEST INFRASTRUCTURE ONLY -- BW6-761 ate pairing in Python big ints, so that the aggregator-shaped proofs
(/root/reference/spec/params/params.go:24, one per batch) get a prove -> verify round trip like the other curves.
Embedding degree 6: Fp6 = Fp[w]/(w^6 + 4); G2 lives over Fp on the M-type sextic twist y^2 = x^3 + 4 of E: y^2 = x^3 - 1 and is
untwisted by (x / w^2, y / w^3).  Plain ate pairing: Miller loop over T = t - 1 (t the trace of E, found from the CM equation as in
tests/test_groth16_curves.py), final exponentiation (p^6 - 1)/r; no sign handling (a fixed power of the pairing is still bilinear
and non-degenerate, which is all a product check needs).  Pinned by bilinearity / order / non-degeneracy only.  Only tests/ import it.
"""
from __future__ import annotations
from math import isqrt

P = 0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B
R = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
D = 6


def _trace():
    """t with r | p + 1 - t and 4p - t^2 = 3 f^2 such that #E(Fp) = p + 1 - t for E: y^2 = x^3 - 1 (checked on a point)."""
    lim = 2 * isqrt(P) + 2
    k = -(lim // R) - 2
    base = []
    while True:
        t = (P + 1) % R + k * R
        if t > lim:
            break
        d = 4 * P - t * t
        if abs(t) <= lim and d > 0 and d % 3 == 0 and 3 * isqrt(d // 3) ** 2 == d:
            base.append((t, isqrt(d // 3)))
        k += 1
    t, f = base[0]
    cands = [t, -t, (t + 3 * f) // 2, (t - 3 * f) // 2, (-t + 3 * f) // 2, (-t - 3 * f) // 2]
    # a point of E: smallest x with x^3 - 1 a square
    x = 2
    while True:
        y2 = (x * x * x - 1) % P
        y = pow(y2, (P + 1) // 4, P)
        if y * y % P == y2:
            break
        x += 1
    def mul(pt, kk):
        acc, b = None, pt
        while kk:
            if kk & 1:
                acc = _aff_add(acc, b)
            b = _aff_add(b, b)
            kk >>= 1
        return acc
    for tt in cands:
        if (P + 1 - tt) % R == 0 and mul((x, y), P + 1 - tt) is None:
            return tt
    raise AssertionError("trace not found")


def _aff_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        m = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        m = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x3 = (m * m - a[0] - b[0]) % P
    return (x3, (m * (a[0] - x3) - a[1]) % P)


class F6:
    """Fp[w]/(w^6 + 4), coefficients little-endian."""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [x % P for x in c]

    @staticmethod
    def one():
        return F6([1] + [0] * (D - 1))

    def __add__(self, o):
        return F6([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return F6([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return F6([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def scale(self, k):
        return F6([a * k for a in self.c])

    def __mul__(self, o):
        t = [0] * (2 * D - 1)
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(o.c):
                    t[i + j] += a * b
        for k in range(2 * D - 2, D - 1, -1):          # w^k = -4 w^(k-6)
            t[k - D] -= 4 * t[k]
        return F6(t[:D])

    def inv(self):
        """a^-1 = a^(p^6 - 2) is too slow in Python; extended Euclid on polynomials over Fp."""
        def deg(p):
            d = len(p) - 1
            while d and p[d] % P == 0:
                d -= 1
            return d

        def poly_div(a, b):
            dega, degb = deg(a), deg(b)
            temp = list(a); o = [0] * len(a)
            for i in range(dega - degb, -1, -1):
                q = temp[degb + i] * pow(b[degb], -1, P) % P
                o[i] = (o[i] + q) % P
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - b[c] * q) % P
            return [x % P for x in o[:deg(o) + 1]]

        n = D + 1
        lm, hm = [1] + [0] * D, [0] * n
        low, high = self.c + [0], [4] + [0] * (D - 1) + [1]
        while deg(low):
            r = poly_div(high, low)
            r += [0] * (n - len(r))
            nm, new = list(hm), list(high)
            for i in range(n):
                for j in range(n - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                    new[i + j] = (new[i + j] - low[i] * r[j]) % P
            lm, low, hm, high = nm, new, lm, low
        li = pow(low[0], -1, P)
        return F6([x * li for x in lm[:D]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e):
        out, base = F6.one(), self
        while e:
            if e & 1:
                out = out * base
            base = base * base
            e >>= 1
        return out


W = F6([0, 1, 0, 0, 0, 0])
_W2I, _W3I = (W * W).inv(), (W * W * W).inv()
T_LOOP = None


def loop_count():
    global T_LOOP
    if T_LOOP is None:
        T_LOOP = abs(_trace() - 1)
    return T_LOOP


def cast(pt):
    return (F6([pt[0]] + [0] * 5), F6([pt[1]] + [0] * 5))


def twist(q):
    """G2 point (x, y) over Fp on y^2 = x^3 + 4 into E(Fp6): (x / w^2, y / w^3)."""
    return (F6([q[0]] + [0] * 5) * _W2I, F6([q[1]] + [0] * 5) * _W3I)


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
    if q2 is None or p1 is None:
        return F6.one()
    Q, Pt = twist(q2), cast(p1)
    n = loop_count()
    Rp, f = Q, F6.one()
    for i in range(n.bit_length() - 2, -1, -1):
        f = f * f * _line(Rp, Rp, Pt)
        Rp = _double(Rp)
        if n & (1 << i):
            f = f * _line(Rp, Q, Pt)
            Rp = _add(Rp, Q)
    return f


def final_exp(f):
    return f ** ((P ** 6 - 1) // R)


def pairing(q2, p1):
    return final_exp(miller_loop(q2, p1))


def pairing_check(pairs) -> bool:
    f = F6.one()
    for p1, q2 in pairs:
        f = f * miller_loop(q2, p1)
    return final_exp(f) == F6.one()
