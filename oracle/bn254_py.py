"""Pure-Python big-integer oracle for the davinci-node prover hot path (BN254).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  The product path is the HIP library behind ``include/dvn_prover.h``.

PARITY UNPINNED: the arithmetic of this path lives in un-vendored Go modules
(``/root/reference/go.mod:15-16`` -- gnark v0.14.1-0.20260126121332-407111efab55 and gnark-crypto
v0.19.3-0.20260126145145-b5cf053fbc34), there is no Go toolchain here, and the reference's own tests hold
no MSM / NTT / proof-byte vectors (every prover test is prove->verify,
``circuits/test/statetransition/statetransition_test.go:89-250``).  What *is* pinned from the reference:
the field/curve constants and on-curve points of ``config/statetransition_vkey.sol:42-114`` and
``circuits/voteverifier/dummy.go:15-47`` (tests/golden/reference_constants.json).  Everything else restates
the published algorithms of those modules as called from ``prover/prover_cpu.go:37``:

* fr/fp ``Element``: 4 x u64 little-endian limbs, Montgomery form, R = 2^256        (gnark-crypto ecc/bn254/fp, fr)
* ``G1Affine`` {X,Y}, infinity = (0,0); ``G2Affine`` over Fp2 = Fp[u]/(u^2+1), element A0 + A1*u
* ``G1Jac.MultiExp``: result is a group element; parity is defined on the affine normal form
* ``fft.Domain.FFT / FFTInverse`` with DIF / DIT and OnCoset, coset shift g = 5       (ecc/bn254/fr/fft)
* groth16 ``computeH`` and the proof assembly Ar / Bs / Krs           (gnark backend/groth16/bn254/prove.go)

This file is written to be trivially auditable, not fast: affine/Jacobian textbook formulas, naive
double-and-add MSM, recursive-free radix-2 loops.
"""
from __future__ import annotations

# --------------------------------------------------------------------------------------------------------
# Constants.  P, R: config/statetransition_vkey.sol:42-43.  Generator (1,2): crypto/ecc/bn254/bn254.go:24-29.
# --------------------------------------------------------------------------------------------------------
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
MONT_R = 1 << 256
B_G1 = 3
G1_GEN = (1, 2)
# EIP-197 / gnark G2 generator, element = (A0, A1) meaning A0 + A1*u
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
FR_MULT_GEN = 5                      # fft.Domain.FrMultiplicativeGen for BN254
FR_TWO_ADICITY = 28
FR_ROOT_2_28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904
assert pow(FR_MULT_GEN, (R - 1) >> FR_TWO_ADICITY, R) == FR_ROOT_2_28
MASK64 = (1 << 64) - 1


# --------------------------------------------------------------------------------------------------------
# Montgomery helpers and limb (de)serialisation
# --------------------------------------------------------------------------------------------------------
def to_mont(x: int, mod: int) -> int:
    return (x << 256) % mod


def from_mont(x: int, mod: int) -> int:
    return (x * pow(MONT_R, -1, mod)) % mod


def limbs(x: int, n: int = 4) -> list[int]:
    return [(x >> (64 * i)) & MASK64 for i in range(n)]


def from_limbs(ls) -> int:
    v = 0
    for i, l in enumerate(ls):
        v |= int(l) << (64 * i)
    return v


def mont_constants(mod: int) -> dict:
    return {
        "R_mod": MONT_R % mod,
        "R2_mod": (MONT_R * MONT_R) % mod,
        "inv64": (-pow(mod, -1, 1 << 64)) % (1 << 64),
        "inv32": (-pow(mod, -1, 1 << 32)) % (1 << 32),
    }


# --------------------------------------------------------------------------------------------------------
# Deterministic input generator shared by the oracle, the C restatement and the device library
# --------------------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def next_fr(self) -> int:
        """4 words little-endian -> 256-bit integer -> reduced mod r (canonical, NOT Montgomery)."""
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R


def splitmix_at(seed: int, index: int) -> int:
    """Counter-mode SplitMix64: the value the ``index``-th call of a generator seeded ``seed`` returns."""
    z = (seed + (index + 1) * 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def fr_at(seed: int, i: int) -> int:
    """i-th canonical Fr scalar of stream ``seed`` (words 4i..4i+3), matching SplitMix64.next_fr order."""
    v = 0
    for k in range(4):
        v |= splitmix_at(seed, 4 * i + k) << (64 * k)
    return v % R


def witness_like_scalar(seed: int, i: int) -> int:
    """'S-witness' mix: 40 % zero, 25 % one, 25 % < 2^64, 10 % uniform (canonical)."""
    sel = splitmix_at(seed ^ 0x5EED5EED5EED5EED, i) % 100
    if sel < 40:
        return 0
    if sel < 65:
        return 1
    if sel < 90:
        return splitmix_at(seed, 4 * i)
    return fr_at(seed, i)


# --------------------------------------------------------------------------------------------------------
# Fields: Fp as int, Fp2 as (a0, a1) with u^2 = -1
# --------------------------------------------------------------------------------------------------------
class Fp:
    zero = 0
    one = 1

    @staticmethod
    def add(a, b): return (a + b) % P
    @staticmethod
    def sub(a, b): return (a - b) % P
    @staticmethod
    def mul(a, b): return (a * b) % P
    @staticmethod
    def neg(a): return (-a) % P
    @staticmethod
    def inv(a): return pow(a, -1, P)
    @staticmethod
    def is_zero(a): return a % P == 0
    @staticmethod
    def from_int(k): return k % P


class Fp2:
    zero = (0, 0)
    one = (1, 0)

    @staticmethod
    def add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    @staticmethod
    def sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
    @staticmethod
    def mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    @staticmethod
    def neg(a): return ((-a[0]) % P, (-a[1]) % P)
    @staticmethod
    def inv(a):
        d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
        return ((a[0] * d) % P, (-a[1] * d) % P)
    @staticmethod
    def is_zero(a): return a[0] % P == 0 and a[1] % P == 0
    @staticmethod
    def from_int(k): return (k % P, 0)


B_G2 = Fp2.mul((3, 0), Fp2.inv((9, 1)))      # twist: y^2 = x^3 + 3/(9+u)


# --------------------------------------------------------------------------------------------------------
# Short Weierstrass y^2 = x^3 + b, a = 0.  Affine points are (x, y) or None (infinity).
# --------------------------------------------------------------------------------------------------------
def ec_is_on_curve(F, b, pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), b)) == F.zero


def ec_neg(F, pt):
    return None if pt is None else (pt[0], F.neg(pt[1]))


def ec_add(F, p1, p2):
    """Textbook affine chord-and-tangent addition (handles doubling, inverses and infinity)."""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if F.is_zero(F.add(y1, y2)):
            return None
        lam = F.mul(F.mul(F.from_int(3), F.mul(x1, x1)), F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


# Jacobian (X, Y, Z) helpers: only used to make the naive scalar multiplication tolerable in Python.
def _jac_double(F, p):
    X, Y, Z = p
    if F.is_zero(Z):
        return p
    A = F.mul(X, X); Bq = F.mul(Y, Y); C = F.mul(Bq, Bq)
    t = F.add(X, Bq); D = F.sub(F.sub(F.mul(t, t), A), C); D = F.add(D, D)
    E = F.add(F.add(A, A), A); Fq = F.mul(E, E)
    X3 = F.sub(Fq, F.add(D, D))
    C8 = F.add(C, C); C8 = F.add(C8, C8); C8 = F.add(C8, C8)
    Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
    Z3 = F.mul(F.add(Y, Y), Z)
    return (X3, Y3, Z3)


def _jac_add_affine(F, p, q):
    """p Jacobian + q affine (q may be None)."""
    if q is None:
        return p
    X1, Y1, Z1 = p
    if F.is_zero(Z1):
        return (q[0], q[1], F.one)
    Z1Z1 = F.mul(Z1, Z1)
    U2 = F.mul(q[0], Z1Z1)
    S2 = F.mul(F.mul(q[1], Z1), Z1Z1)
    H = F.sub(U2, X1)
    r = F.sub(S2, Y1)
    if F.is_zero(H):
        if F.is_zero(r):
            return _jac_double(F, p)
        return (F.one, F.one, F.zero)
    HH = F.mul(H, H); HHH = F.mul(H, HH); V = F.mul(X1, HH)
    X3 = F.sub(F.sub(F.mul(r, r), HHH), F.add(V, V))
    Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.mul(Y1, HHH))
    Z3 = F.mul(Z1, H)
    return (X3, Y3, Z3)


def _jac_to_affine(F, p):
    X, Y, Z = p
    if F.is_zero(Z):
        return None
    zi = F.inv(Z); zi2 = F.mul(zi, zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


def ec_mul(F, pt, k: int):
    """Naive MSB-first double-and-add of an affine point by a non-negative integer."""
    if pt is None or k == 0:
        return None
    acc = (F.one, F.one, F.zero)
    for bit in bin(k)[2:]:
        acc = _jac_double(F, acc)
        if bit == "1":
            acc = _jac_add_affine(F, acc, pt)
    return _jac_to_affine(F, acc)


def ec_msm(F, points, scalars):
    """Sum_i scalars[i] * points[i], the definition gnark-crypto's MultiExp must agree with."""
    if len(points) != len(scalars):
        raise ValueError("len(points) != len(scalars)")       # same error condition as MultiExp
    acc = None
    for pt, k in zip(points, scalars):
        acc = ec_add(F, acc, ec_mul(F, pt, k % R))
    return acc


# Convenience wrappers --------------------------------------------------------------------------------------
def g1_add(a, b): return ec_add(Fp, a, b)
def g1_neg(a): return ec_neg(Fp, a)
def g1_mul(a, k): return ec_mul(Fp, a, k)
def g1_on_curve(a): return ec_is_on_curve(Fp, B_G1, a)
def g1_msm(pts, ks): return ec_msm(Fp, pts, ks)
def g2_add(a, b): return ec_add(Fp2, a, b)
def g2_neg(a): return ec_neg(Fp2, a)
def g2_mul(a, k): return ec_mul(Fp2, a, k)
def g2_on_curve(a): return ec_is_on_curve(Fp2, B_G2, a)
def g2_msm(pts, ks): return ec_msm(Fp2, pts, ks)


def g1_to_limbs(pt) -> list[int]:
    """gnark G1Affine memory image: X then Y, Montgomery limbs; infinity = (0,0)."""
    if pt is None:
        return [0] * 8
    return limbs(to_mont(pt[0], P)) + limbs(to_mont(pt[1], P))


def g1_from_limbs(ls):
    x = from_mont(from_limbs(ls[0:4]), P); y = from_mont(from_limbs(ls[4:8]), P)
    return None if (x == 0 and y == 0) else (x, y)


def g2_to_limbs(pt) -> list[int]:
    """gnark G2Affine memory image: X.A0, X.A1, Y.A0, Y.A1."""
    if pt is None:
        return [0] * 16
    out = []
    for c in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]):
        out += limbs(to_mont(c, P))
    return out


def g2_from_limbs(ls):
    c = [from_mont(from_limbs(ls[4 * i:4 * i + 4]), P) for i in range(4)]
    return None if all(v == 0 for v in c) else ((c[0], c[1]), (c[2], c[3]))


def synthetic_g1_points(seed: int, n: int):
    """P_i = P_0 + i*D with P_0 = [s0]G, D = [s1]G, (s0, s1) the first two Fr of stream ``seed``."""
    s0, s1 = fr_at(seed, 0), fr_at(seed, 1)
    p0, d = g1_mul(G1_GEN, s0), g1_mul(G1_GEN, s1)
    out, cur = [], p0
    for _ in range(n):
        out.append(cur)
        cur = g1_add(cur, d)
    return out


def synthetic_g2_points(seed: int, n: int):
    s0, s1 = fr_at(seed, 0), fr_at(seed, 1)
    p0, d = g2_mul(G2_GEN, s0), g2_mul(G2_GEN, s1)
    out, cur = [], p0
    for _ in range(n):
        out.append(cur)
        cur = g2_add(cur, d)
    return out


# --------------------------------------------------------------------------------------------------------
# fr/fft Domain.  All vectors are canonical integers mod R here; the Montgomery image is a
# serialisation detail handled by to_mont/limbs.
# --------------------------------------------------------------------------------------------------------
def bitrev(i: int, log_n: int) -> int:
    return int(format(i, "0%db" % log_n)[::-1], 2) if log_n else 0


DIF, DIT = 0, 1


class Domain:
    def __init__(self, m: int):
        n = 1
        while n < m:
            n <<= 1
        self.n = n
        self.log_n = n.bit_length() - 1
        assert self.log_n <= FR_TWO_ADICITY
        self.gen = pow(FR_ROOT_2_28, 1 << (FR_TWO_ADICITY - self.log_n), R)
        self.gen_inv = pow(self.gen, -1, R)
        self.card_inv = pow(n, -1, R)
        self.coset = FR_MULT_GEN
        self.coset_inv = pow(FR_MULT_GEN, -1, R)

    # butterflies exactly as gnark-crypto's fft.Domain orders them
    def _dif(self, a, w):
        n = self.n
        half = n >> 1
        stride = 1          # twiddle exponent multiplier
        while half >= 1:
            for start in range(0, n, 2 * half):
                for j in range(half):
                    x, y = a[start + j], a[start + j + half]
                    a[start + j] = (x + y) % R
                    a[start + j + half] = ((x - y) * pow(w, j * stride, R)) % R
            half >>= 1
            stride <<= 1

    def _dit(self, a, w):
        n = self.n
        half = 1
        stride = n >> 1
        while half < n:
            for start in range(0, n, 2 * half):
                for j in range(half):
                    x = a[start + j]
                    y = (a[start + j + half] * pow(w, j * stride, R)) % R
                    a[start + j] = (x + y) % R
                    a[start + j + half] = (x - y) % R
            half <<= 1
            stride >>= 1

    def fft(self, a, decimation, coset=False):
        assert len(a) == self.n
        if coset:
            for i in range(self.n):
                k = bitrev(i, self.log_n) if decimation == DIT else i
                a[i] = (a[i] * pow(self.coset, k, R)) % R
        (self._dif if decimation == DIF else self._dit)(a, self.gen)

    def fft_inverse(self, a, decimation, coset=False):
        assert len(a) == self.n
        (self._dif if decimation == DIF else self._dit)(a, self.gen_inv)
        for i in range(self.n):
            s = self.card_inv
            if coset:
                k = bitrev(i, self.log_n) if decimation == DIF else i
                s = (s * pow(self.coset_inv, k, R)) % R
            a[i] = (a[i] * s) % R


def naive_dft(a, w):
    n = len(a)
    return [sum(a[j] * pow(w, i * j, R) for j in range(n)) % R for i in range(n)]


def compute_h(a, b, c, domain: Domain):
    """gnark groth16 computeH.  Returns n values, coefficients of h in bit-reversed order."""
    n = domain.n
    a = list(a) + [0] * (n - len(a))
    b = list(b) + [0] * (n - len(b))
    c = list(c) + [0] * (n - len(c))
    for v in (a, b, c):
        domain.fft_inverse(v, DIF)
    for v in (a, b, c):
        domain.fft(v, DIT, coset=True)
    den = pow((pow(domain.coset, n, R) - 1) % R, -1, R)
    for i in range(n):
        a[i] = ((a[i] * b[i] - c[i]) * den) % R
    domain.fft_inverse(a, DIF, coset=True)
    return a


# --------------------------------------------------------------------------------------------------------
# Synthetic Groth16 instance with a known trapdoor
# --------------------------------------------------------------------------------------------------------
class SyntheticR1CS:
    """n_c constraints (w[l] + k*w[l2]) * w[r] = w[o]; wire 0 is the constant 1, wires 1..n_pub-1 public.

    Rows are sparse lists of (wire, coeff).  The witness is built forward so every constraint holds.
    """

    def __init__(self, n_constraints: int, n_public: int, seed: int):
        assert n_public >= 1
        rng = SplitMix64(seed)
        self.n_public = n_public
        self.n_constraints = n_constraints
        w = [1] + [rng.next_fr() for _ in range(n_public - 1)]
        # a few free private inputs so early gates have operands
        n_free = 4
        w += [rng.next_fr() for _ in range(n_free)]
        self.A, self.B, self.C = [], [], []
        for _ in range(n_constraints):
            m = len(w)
            l1 = rng.next() % m
            l2 = rng.next() % m
            k = rng.next() % 7 + 1
            r_ = rng.next() % m
            o = m
            val = ((w[l1] + k * w[l2]) * w[r_]) % R
            w.append(val)
            self.A.append([(l1, 1), (l2, k)] if l1 != l2 else [(l1, (1 + k) % R)])
            self.B.append([(r_, 1)])
            self.C.append([(o, 1)])
        self.witness = w
        self.n_wires = len(w)

    def abc(self):
        def ev(rows):
            return [sum(c * self.witness[i] for i, c in row) % R for row in rows]
        return ev(self.A), ev(self.B), ev(self.C)


def lagrange_at(domain: Domain, tau: int):
    """L_j(tau) for j = 0..n-1 over the domain {gen^j}."""
    n = domain.n
    zt = (pow(tau, n, R) - 1) % R
    # zt w^j / (n (tau - w^j)) with the n denominators inverted together (Montgomery's trick, one modular inversion): the
    # closed-form checker then runs at the benchmarked sizes (n = 2^20) in seconds
    ws, den, wj = [], [], 1
    for _ in range(n):
        ws.append(wj); den.append(n * (tau - wj) % R); wj = wj * domain.gen % R
    assert all(den), "tau lies in the domain"
    pre, acc = [], 1
    for d in den:
        pre.append(acc); acc = acc * d % R
    inv = pow(acc, -1, R)
    out = [0] * n
    for j in range(n - 1, -1, -1):
        out[j] = zt * ws[j] % R * (inv * pre[j] % R) % R
        inv = inv * den[j] % R
    return out


def groth16_setup(cs: SyntheticR1CS, seed: int):
    """Returns (pk, trapdoor).  pk mirrors gnark's bn254 ProvingKey fields used by Prove:
    G1.{Alpha,Beta,Delta,A,B,K,Z}, G2.{Beta,Delta,B}.  No infinity filtering (InfinityA/B empty)."""
    rng = SplitMix64(seed)
    tau, alpha, beta, gamma, delta = (rng.next_fr() or 1 for _ in range(5))
    dom = Domain(cs.n_constraints)
    n = dom.n
    L = lagrange_at(dom, tau)
    m = cs.n_wires
    At, Bt, Ct = [0] * m, [0] * m, [0] * m
    for rows, acc in ((cs.A, At), (cs.B, Bt), (cs.C, Ct)):
        for j, row in enumerate(rows):
            for i, c in row:
                acc[i] = (acc[i] + c * L[j]) % R
    dinv = pow(delta, -1, R)
    K = [((beta * At[i] + alpha * Bt[i] + Ct[i]) % R) * dinv % R for i in range(cs.n_public, m)]
    zt = (pow(tau, n, R) - 1) * dinv % R
    Zs = [zt * pow(tau, i, R) % R for i in range(n)]
    Zs_br = [Zs[bitrev(i, dom.log_n)] for i in range(n)]       # gnark bit-reverses pk.G1.Z at setup
    pk = {
        "domain": dom,
        "G1": {
            "Alpha": g1_mul(G1_GEN, alpha), "Beta": g1_mul(G1_GEN, beta), "Delta": g1_mul(G1_GEN, delta),
            "A": [g1_mul(G1_GEN, v) for v in At], "B": [g1_mul(G1_GEN, v) for v in Bt],
            "K": [g1_mul(G1_GEN, v) for v in K], "Z": [g1_mul(G1_GEN, v) for v in Zs_br],
        },
        "G2": {
            "Beta": g2_mul(G2_GEN, beta), "Delta": g2_mul(G2_GEN, delta),
            "B": [g2_mul(G2_GEN, v) for v in Bt],
        },
    }
    trap = {"tau": tau, "alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta,
            "At": At, "Bt": Bt, "Ct": Ct, "K": K}
    return pk, trap


def groth16_prove(cs: SyntheticR1CS, pk, r: int, s: int):
    """gnark groth16 Prove after the solver (computeH, the five MSMs, blinding) with injected blinding (r, s)."""
    dom = pk["domain"]
    a, b, c = cs.abc()
    h = compute_h(a, b, c, dom)
    w = cs.witness
    n = dom.n
    g1 = pk["G1"]
    ar = g1_add(g1_add(g1_msm(g1["A"], w), g1["Alpha"]), g1_mul(g1["Delta"], r))
    bs1 = g1_add(g1_add(g1_msm(g1["B"], w), g1["Beta"]), g1_mul(g1["Delta"], s))
    bs = g2_add(g2_add(g2_msm(pk["G2"]["B"], w), pk["G2"]["Beta"]), g2_mul(pk["G2"]["Delta"], s))
    krs = g1_msm(g1["K"], w[cs.n_public:])
    krs = g1_add(krs, g1_msm(g1["Z"][: n - 1], h[: n - 1]))
    krs = g1_add(krs, g1_mul(g1["Delta"], (-r * s) % R))
    krs = g1_add(krs, g1_mul(ar, s))
    krs = g1_add(krs, g1_mul(bs1, r))
    return {"Ar": ar, "Bs": bs, "Krs": krs, "h": h}


def groth16_expected_from_trapdoor(cs: SyntheticR1CS, trap, dom: Domain, r: int, s: int):
    """Closed-form proof from the trapdoor: O(n) field work + 3 fixed-base scalar multiplications."""
    w = cs.witness
    tau, alpha, beta, delta = trap["tau"], trap["alpha"], trap["beta"], trap["delta"]
    a_tau = sum(wi * v for wi, v in zip(w, trap["At"])) % R
    b_tau = sum(wi * v for wi, v in zip(w, trap["Bt"])) % R
    c_tau = sum(wi * v for wi, v in zip(w, trap["Ct"])) % R
    n = dom.n
    z_tau = (pow(tau, n, R) - 1) % R
    h_tau = (a_tau * b_tau - c_tau) * pow(z_tau, -1, R) % R
    ar = (alpha + a_tau + r * delta) % R
    bs = (beta + b_tau + s * delta) % R
    priv = sum(w[cs.n_public + i] * k for i, k in enumerate(trap["K"])) % R
    krs = (priv + h_tau * z_tau % R * pow(delta, -1, R) + s * ar + r * bs - r * s % R * delta) % R
    return {"Ar": g1_mul(G1_GEN, ar), "Bs": g2_mul(G2_GEN, bs), "Krs": g1_mul(G1_GEN, krs)}


# ------------------------------------------------------------------------------------------------------
# BSB22 commitment: the statetransition circuit has exactly one
# (/root/reference/solidity/solidity.go:57-65; 8 public inputs + 1 commitment wire = PUB_0..PUB_8,
# /root/reference/config/statetransition_vkey.sol:97-114).  What the reference tree itself pins:
#   challenge  = keccak256(C.x || C.y || publicCommitted...) mod r            statetransition_vkey.sol:659-676
#   vk_x       = CONSTANT + C + sum_i input_i PUB_i + challenge PUB_8         statetransition_vkey.sol:393-520
#   PoK check  e(C, GSigmaNeg) e(pok, G) = 1                                  statetransition_vkey.sol:678-700
# so C = [sum_committed w_i (beta A_i + alpha B_i + C_i)(tau) / gamma]G1 (it is added to the gamma-paired accumulator)
# and pok = sigma C for the sigma with GSigmaNeg = -sigma G.
# ------------------------------------------------------------------------------------------------------
_KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
              0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
              0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
              0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KECCAK_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]


def _keccak_f(st):
    M = (1 << 64) - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & M if n else v
    for rc in _KECCAK_RC:
        c = [st[x][0] ^ st[x][1] ^ st[x][2] ^ st[x][3] ^ st[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        st = [[st[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = rol(st[x][y], _KECCAK_ROT[x][y])
        st = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        st[0][0] ^= rc
    return st


def keccak256(data: bytes) -> bytes:
    """Legacy Keccak-256 (Ethereum's; pad 0x01 .. 0x80, rate 136), what Solidity's keccak256 computes."""
    rate = 136
    p = bytearray(data) + b"\x01" + bytes((-len(data) - 2) % rate) + b"\x80" if (len(data) + 1) % rate else bytearray(data) + b"\x81"
    st = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        for i in range(rate // 8):
            st[i % 5][i // 5] ^= int.from_bytes(p[off + 8 * i:off + 8 * i + 8], "little")
        st = _keccak_f(st)
    return b"".join(st[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class SyntheticR1CSCommitted:
    """SyntheticR1CS with ONE BSB22 commitment.  Wire order as gnark's: [1, public inputs, commitment wire (public
    section), private ...]; the first n_committed private wires are the committed ones; public_committed lists the
    public wires hashed into the challenge (statetransition: the third input, vkey.sol:665).  Only the last `tail`
    constraints read the commitment wire, so completing the witness once the commitment is known is O(tail)."""

    def __init__(self, n_constraints: int, n_public: int, n_committed: int, seed: int, tail: int = 2):
        assert n_public >= 1 and n_constraints > tail and n_committed >= 1
        rng = SplitMix64(seed)
        self.n_inputs = n_public                       # constant + public inputs
        self.commit_wire = n_public
        self.n_public = n_public + 1                   # public section incl. the commitment wire
        self.n_committed, self.tail, self.n_constraints = n_committed, tail, n_constraints
        self.committed = list(range(self.n_public, self.n_public + n_committed))
        self.public_committed = [3] if n_public >= 4 else ([n_public - 1] if n_public >= 2 else [])
        w = [1] + [rng.next_fr() for _ in range(n_public - 1)] + [0]
        w += [rng.next_fr() for _ in range(4 + n_committed)]
        self.A, self.B, self.C = [], [], []
        cw = self.commit_wire
        for j in range(n_constraints):
            m = len(w)
            l1, l2 = rng.next() % m, rng.next() % m
            k = rng.next() % 7 + 1
            r_ = rng.next() % m
            l2 = 0 if l2 == cw else l2
            r_ = 0 if r_ == cw else r_
            l1 = cw if j >= n_constraints - tail else (0 if l1 == cw else l1)
            w.append(((w[l1] + k * w[l2]) * w[r_]) % R)
            self.A.append([(l1, 1), (l2, k)] if l1 != l2 else [(l1, (1 + k) % R)])
            self.B.append([(r_, 1)])
            self.C.append([(m, 1)])
        self.witness = w                               # commitment wire = 0, tail outputs computed with it: incomplete
        self.n_wires = len(w)

    def complete(self, challenge: int):
        """The solver's last step: commitment wire := challenge, then the constraints that read it, in order."""
        w = list(self.witness)
        w[self.commit_wire] = challenge % R
        for j in range(self.n_constraints - self.tail, self.n_constraints):
            a = sum(c * w[i] for i, c in self.A[j]) % R
            b = sum(c * w[i] for i, c in self.B[j]) % R
            w[self.C[j][0][0]] = a * b % R
        return w


def commitment_challenge(cpt, public_values):
    """statetransition_vkey.sol:659-676: keccak256(abi.encodePacked(C.x, C.y, publicAndCommitmentCommitted)) % R."""
    x, y = (0, 0) if cpt is None else cpt
    data = x.to_bytes(32, "big") + y.to_bytes(32, "big") + b"".join(v.to_bytes(32, "big") for v in public_values)
    return int.from_bytes(keccak256(data), "big") % R


def groth16_committed_expected(cs: SyntheticR1CSCommitted, seed: int, r: int, s: int):
    """Closed-form proof with one commitment from the trapdoor (tau, alpha, beta, gamma, delta, sigma drawn in that order
    from SplitMix64(seed)): O(n) field work + 5 fixed-base multiplications.  Returns the proof points, the completed
    witness and the exponents (so a test can check the verifier's equation in the exponent)."""
    rng = SplitMix64(seed)
    tau, alpha, beta, gamma, delta, sigma = (rng.next_fr() or 1 for _ in range(6))
    dom = Domain(cs.n_constraints)
    n = dom.n
    L = lagrange_at(dom, tau)
    m = cs.n_wires
    At, Bt, Ct = [0] * m, [0] * m, [0] * m
    for rows, acc in ((cs.A, At), (cs.B, Bt), (cs.C, Ct)):
        for j, row in enumerate(rows):
            for i, c in row:
                acc[i] = (acc[i] + c * L[j]) % R
    t = [(beta * At[i] + alpha * Bt[i] + Ct[i]) % R for i in range(m)]
    ginv, dinv = pow(gamma, -1, R), pow(delta, -1, R)
    cexp = sum(cs.witness[i] * t[i] for i in cs.committed) % R * ginv % R
    cpt = g1_mul(G1_GEN, cexp)
    ch = commitment_challenge(cpt, [cs.witness[i] for i in cs.public_committed])
    w = cs.complete(ch)
    a_tau = sum(wi * v for wi, v in zip(w, At)) % R
    b_tau = sum(wi * v for wi, v in zip(w, Bt)) % R
    c_tau = sum(wi * v for wi, v in zip(w, Ct)) % R
    z_tau = (pow(tau, n, R) - 1) % R
    h_tau = (a_tau * b_tau - c_tau) * pow(z_tau, -1, R) % R
    ar = (alpha + a_tau + r * delta) % R
    bs = (beta + b_tau + s * delta) % R
    first_k = cs.n_public + cs.n_committed
    priv = sum(w[i] * t[i] for i in range(first_k, m)) % R * dinv % R
    krs = (priv + h_tau * z_tau % R * dinv + s * ar + r * bs - r * s % R * delta) % R
    vkx = (sum(w[i] * t[i] for i in range(cs.n_public)) % R * ginv + cexp) % R     # CONSTANT + inputs + challenge wire, + C
    # the verifying key in the shape of the constants of config/statetransition_vkey.sol:62-114
    vk = {"ALPHA": g1_mul(G1_GEN, alpha), "BETA_NEG": g2_neg(g2_mul(G2_GEN, beta)), "GAMMA_NEG": g2_neg(g2_mul(G2_GEN, gamma)),
          "DELTA_NEG": g2_neg(g2_mul(G2_GEN, delta)), "CONSTANT": g1_mul(G1_GEN, t[0] * ginv % R),
          "PUB": [g1_mul(G1_GEN, t[i] * ginv % R) for i in range(1, cs.n_public)],
          "PEDERSEN_G": G2_GEN, "PEDERSEN_GSIGMANEG": g2_neg(g2_mul(G2_GEN, sigma)),
          "PUBLIC_COMMITTED": [i - 1 for i in cs.public_committed]}
    return {"Ar": g1_mul(G1_GEN, ar), "Bs": g2_mul(G2_GEN, bs), "Krs": g1_mul(G1_GEN, krs), "Commitment": cpt,
            "CommitmentPok": g1_mul(G1_GEN, sigma * cexp % R), "challenge": ch, "witness": w, "vk": vk,
            "exponents": {"ar": ar, "bs": bs, "krs": krs, "vkx": vkx, "alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta,
                          "sigma": sigma, "cexp": cexp}}
