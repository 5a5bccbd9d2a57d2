"""Curve-generic big-integer oracle (TEST INFRASTRUCTURE ONLY; see oracle/bn254_py.py header for the rules).

Extends the BN254 oracle to the other curves the reference proves on: BLS12-377
(voteverifier, `/root/reference/spec/params/params.go:21-27`; GPU dispatch at `prover/prover_gpu.go:36-43`).
Same conventions as gnark-crypto: Element = little-endian u64 limbs in Montgomery form with R = 2^(64 * limbs)
(4 limbs for the 253/254-bit fields, 6 for BLS12-377's 377-bit base field); E2 = a0 + a1*u with u^2 = -BETA;
affine infinity = all-zero image; fft.Domain DIF/DIT/coset semantics with the curve's multiplicative generator.

BW6-761 (aggregator, params.go:24) is the `Bw6761` class at the bottom: both groups over the 761-bit base field.

PARITY UNPINNED: no vectors for these primitives exist in the reference.  The constants below were cross-checked here:
r | #E(Fp) via r*G1 = O, G1 on y^2 = x^3 + 1, root_of_unity == gen^((r-1)/2^47) == gnark-crypto's hard-coded
8065159656716812877374967518403273466521432693661810619979959746626482506078, twist coefficient == 1/u = (0, -1/5).
"""
from __future__ import annotations
from . import bn254_py as b

MASK64 = (1 << 64) - 1


class PrimeField:
    def __init__(self, p):
        self.p = p
        self.zero, self.one = 0, 1

    def add(self, a, c): return (a + c) % self.p
    def sub(self, a, c): return (a - c) % self.p
    def mul(self, a, c): return (a * c) % self.p
    def neg(self, a): return (-a) % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def is_zero(self, a): return a % self.p == 0
    def from_int(self, k): return k % self.p


class QuadExt:
    """Fp[u]/(u^2 + beta), elements (a0, a1)."""

    def __init__(self, p, beta):
        self.p, self.beta = p, beta
        self.zero, self.one = (0, 0), (1, 0)

    def add(self, a, c): return ((a[0] + c[0]) % self.p, (a[1] + c[1]) % self.p)
    def sub(self, a, c): return ((a[0] - c[0]) % self.p, (a[1] - c[1]) % self.p)
    def mul(self, a, c):
        return ((a[0] * c[0] - self.beta * a[1] * c[1]) % self.p, (a[0] * c[1] + a[1] * c[0]) % self.p)
    def neg(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)
    def inv(self, a):
        d = pow(a[0] * a[0] + self.beta * a[1] * a[1], -1, self.p)
        return ((a[0] * d) % self.p, (-a[1] * d) % self.p)
    def is_zero(self, a): return a[0] % self.p == 0 and a[1] % self.p == 0
    def from_int(self, k): return (k % self.p, 0)


def _tonelli(n, p):
    """Square root mod p (p odd prime), or None."""
    n %= p
    if n == 0:
        return 0
    if pow(n, (p - 1) // 2, p) != 1:
        return None
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2; s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r_ = s, pow(z, q, p), pow(n, q, p), pow(n, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p; i += 1
        bb = pow(c, 1 << (m - i - 1), p)
        m, c = i, bb * bb % p
        t, r_ = t * c % p, r_ * bb % p
    return r_


class Suite:
    """One pairing-friendly curve: base/scalar fields, G1: y^2 = x^3 + b, G2 over QuadExt: y^2 = x^3 + b2."""

    def __init__(self, name, p, r, b_g1, beta, b_g2, g1_gen, fr_gen, two_adicity):
        self.name, self.p, self.r = name, p, r
        self.fp, self.fp2 = PrimeField(p), QuadExt(p, beta)
        self.b_g1, self.b_g2, self.g1_gen = b_g1, b_g2, g1_gen
        self.fp_limbs = (p.bit_length() + 63) // 64
        self.fr_limbs = (r.bit_length() + 63) // 64
        self.fr_gen, self.two_adicity = fr_gen, two_adicity
        self.root = pow(fr_gen, (r - 1) >> two_adicity, r)

    # ---- Montgomery images -------------------------------------------------------------------------------------
    def fp_to_mont(self, x): return (x << (64 * self.fp_limbs)) % self.p
    def fp_from_mont(self, x): return x * pow(1 << (64 * self.fp_limbs), -1, self.p) % self.p
    def fr_to_mont(self, x): return (x << (64 * self.fr_limbs)) % self.r
    def fr_from_mont(self, x): return x * pow(1 << (64 * self.fr_limbs), -1, self.r) % self.r

    def limbs(self, x, n): return [(x >> (64 * i)) & MASK64 for i in range(n)]

    def g1_to_limbs(self, pt):
        n = self.fp_limbs
        if pt is None:
            return [0] * (2 * n)
        return self.limbs(self.fp_to_mont(pt[0]), n) + self.limbs(self.fp_to_mont(pt[1]), n)

    def g2_to_limbs(self, pt):
        n = self.fp_limbs
        if pt is None:
            return [0] * (4 * n)
        out = []
        for c in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]):
            out += self.limbs(self.fp_to_mont(c), n)
        return out

    def fr_to_limbs(self, x): return self.limbs(self.fr_to_mont(x), self.fr_limbs)
    def fr_from_limbs(self, ls): return self.fr_from_mont(b.from_limbs(ls))

    # ---- groups (generic chord-and-tangent law of bn254_py) ----------------------------------------------------------
    def g1_on_curve(self, pt): return b.ec_is_on_curve(self.fp, self.b_g1, pt)
    def g2_on_curve(self, pt): return b.ec_is_on_curve(self.fp2, self.b_g2, pt)
    def g1_add(self, a, c): return b.ec_add(self.fp, a, c)
    def g2_add(self, a, c): return b.ec_add(self.fp2, a, c)
    def g1_neg(self, a): return b.ec_neg(self.fp, a)
    def g2_neg(self, a): return b.ec_neg(self.fp2, a)
    def g1_mul(self, a, k): return b.ec_mul(self.fp, a, k)
    def g2_mul(self, a, k): return b.ec_mul(self.fp2, a, k)

    def msm(self, F, pts, ks):
        if len(pts) != len(ks):
            raise ValueError("len(points) != len(scalars)")
        acc = None
        for pt, k in zip(pts, ks):
            acc = b.ec_add(F, acc, b.ec_mul(F, pt, k % self.r))
        return acc

    def g1_msm(self, pts, ks): return self.msm(self.fp, pts, ks)
    def g2_msm(self, pts, ks): return self.msm(self.fp2, pts, ks)

    def fr_at(self, seed, i):
        """i-th canonical scalar of SplitMix64 stream `seed` (256-bit value mod r), as bn254_py.fr_at."""
        v = 0
        for k in range(4):
            v |= b.splitmix_at(seed, 4 * i + k) << (64 * k)
        return v % self.r

    def synthetic_g1_points(self, seed, n):
        s0, s1 = self.fr_at(seed, 0), self.fr_at(seed, 1)
        p0, d = self.g1_mul(self.g1_gen, s0), self.g1_mul(self.g1_gen, s1)
        out, cur = [], p0
        for _ in range(n):
            out.append(cur); cur = self.g1_add(cur, d)
        return out

    def fp2_sqrt(self, a):
        """sqrt in Fp[u]/(u^2+beta) by the norm method; None if a is not a square."""
        p, beta = self.p, self.fp2.beta
        a0, a1 = a
        if a1 % p == 0:
            s = _tonelli(a0, p)
            if s is not None:
                return (s, 0)
            s = _tonelli(a0 * pow(-beta, -1, p) % p, p)        # a0 = -beta * t^2  ->  sqrt = t*u
            return None if s is None else (0, s)
        n = _tonelli((a0 * a0 + beta * a1 * a1) % p, p)
        if n is None:
            return None
        for nn in (n, (-n) % p):
            x2 = (a0 + nn) * pow(2, -1, p) % p
            x = _tonelli(x2, p)
            if x is not None and x != 0:
                y = a1 * pow(2 * x, -1, p) % p
                if self.fp2.mul((x, y), (x, y)) == (a0 % p, a1 % p):
                    return (x, y)
        return None

    def g2_some_point(self, start=1):
        """A point on the twist (not necessarily in the r-torsion: the group law under test does not care)."""
        x0 = start
        while True:
            x = (x0, 1)
            rhs = self.fp2.add(self.fp2.mul(self.fp2.mul(x, x), x), self.b_g2)
            y = self.fp2_sqrt(rhs)
            if y is not None:
                return (x, y)
            x0 += 1

    def synthetic_g2_points(self, seed, n):
        base = self.g2_some_point(1 + seed % 97)
        s0, s1 = self.fr_at(seed, 0), self.fr_at(seed, 1)
        p0, d = self.g2_mul(base, s0), self.g2_mul(base, s1)
        out, cur = [], p0
        for _ in range(n):
            out.append(cur); cur = self.g2_add(cur, d)
        return out

    # ---- fr/fft Domain with gnark semantics (bn254_py.Domain restated for any scalar field) ---------------------------
    def domain(self, m):
        return GenericDomain(self, m)


class GenericDomain:
    def __init__(self, suite: Suite, m: int):
        n = 1
        while n < m:
            n <<= 1
        self.n, self.log_n, self.r = n, n.bit_length() - 1, suite.r
        assert self.log_n <= suite.two_adicity
        self.gen = pow(suite.root, 1 << (suite.two_adicity - self.log_n), self.r)
        self.gen_inv = pow(self.gen, -1, self.r)
        self.card_inv = pow(n, -1, self.r)
        self.coset, self.coset_inv = suite.fr_gen, pow(suite.fr_gen, -1, self.r)

    def _dif(self, a, w):
        n, R = self.n, self.r
        half, stride = n >> 1, 1
        while half >= 1:
            for start in range(0, n, 2 * half):
                for j in range(half):
                    x, y = a[start + j], a[start + j + half]
                    a[start + j] = (x + y) % R
                    a[start + j + half] = ((x - y) * pow(w, j * stride, R)) % R
            half >>= 1; stride <<= 1

    def _dit(self, a, w):
        n, R = self.n, self.r
        half, stride = 1, n >> 1
        while half < n:
            for start in range(0, n, 2 * half):
                for j in range(half):
                    x = a[start + j]
                    y = (a[start + j + half] * pow(w, j * stride, R)) % R
                    a[start + j] = (x + y) % R
                    a[start + j + half] = (x - y) % R
            half <<= 1; stride >>= 1

    def fft(self, a, decimation, coset=False):
        if coset:
            for i in range(self.n):
                k = b.bitrev(i, self.log_n) if decimation == b.DIT else i
                a[i] = (a[i] * pow(self.coset, k, self.r)) % self.r
        (self._dif if decimation == b.DIF else self._dit)(a, self.gen)

    def fft_inverse(self, a, decimation, coset=False):
        (self._dif if decimation == b.DIF else self._dit)(a, self.gen_inv)
        for i in range(self.n):
            s = self.card_inv
            if coset:
                k = b.bitrev(i, self.log_n) if decimation == b.DIF else i
                s = (s * pow(self.coset_inv, k, self.r)) % self.r
            a[i] = (a[i] * s) % self.r

    def compute_h(self, a, bb, c):
        """gnark groth16 computeH on this domain (bn254_py.compute_h, any scalar field): n values, bit-reversed."""
        n, R = self.n, self.r
        a = list(a) + [0] * (n - len(a)); bb = list(bb) + [0] * (n - len(bb)); c = list(c) + [0] * (n - len(c))
        for v in (a, bb, c):
            self.fft_inverse(v, b.DIF)
        for v in (a, bb, c):
            self.fft(v, b.DIT, coset=True)
        den = pow((pow(self.coset, n, R) - 1) % R, -1, R)
        for i in range(n):
            a[i] = ((a[i] * bb[i] - c[i]) * den) % R
        self.fft_inverse(a, b.DIF, coset=True)
        return a


_P377 = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
_R377 = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
BLS12_377 = Suite(
    "bls12-377", _P377, _R377, b_g1=1, beta=5, b_g2=(0, (-pow(5, -1, _P377)) % _P377),
    g1_gen=(0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
            0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6),
    fr_gen=22, two_adicity=47)
# r-torsion generator of the BLS12-377 twist used by the library's SYNTHETIC Groth16 setup: [#E'(Fp2)/r] * (smallest-x twist point);
# not gnark's generator (tests/test_groth16_curves.py re-derives it from the trace)
BLS12_377.g2_gen = (
    (0x6f72205595a839df693176b247c2fa251f7e02a29061e50540dc9e1c2bf1957bf1bab2288c257c2cb36b58f2418bc9,
     0x138c24b2b4e17888beed0a9802aac837cdea39890effe00072f754ecb0152dd6cb524f281298966dbaeca23d3e462b8),
    (0x16235fdea6c3faf2a83d3730f6ab2c033ef6c2739002946f7dc48e4688bca1af1c9b417d58220817e0dc644b5e7d916,
     0x707ac6cc7d192827fc54eb83267f3bed8511bd3c74f63a1ea75eabb66476769c8786f2af2a75166f33142379b4963c))

_P381 = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_R381 = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
# BLS12-381 (blob KZG, prover_gpu.go:45-50): y^2 = x^3 + 4, twist y^2 = x^3 + 4(1 + u), u^2 = -1, Fr generator 7
BLS12_381 = Suite(
    "bls12-381", _P381, _R381, b_g1=4, beta=1, b_g2=(4, 4),
    g1_gen=(0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
            0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
    fr_gen=7, two_adicity=32)
BLS12_381.g2_gen = (      # the standard generator of G2 (r-torsion checked in tests/test_groth16_curves.py)
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))
BN254 = Suite("bn254", b.P, b.R, b_g1=3, beta=1, b_g2=b.B_G2, g1_gen=b.G1_GEN, fr_gen=5, two_adicity=28)


class Bw6761:
    """BW6-761 (gnark-crypto ecc/bw6-761): G1: y^2 = x^3 - 1 and G2: y^2 = x^3 + 4, both over the 761-bit base field;
    the scalar field is BLS12-377's base field (377 bits).  Element images: 12 x u64 (R = 2^768) for coordinates,
    6 x u64 (R = 2^384) for scalars.  Constants cross-checked in tests/test_bw6761.py (primality, embedding degree 6,
    p = 3 mod 4, r | #E for both curves through cofactor-cleared points)."""
    p = 0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B
    r = _P377
    fp_limbs, fr_limbs = 12, 6
    name = "bw6-761"

    def __init__(self):
        self.fp = PrimeField(self.p)
        self.b = {1: self.p - 1, 2: 4}

    def fp_to_mont(self, x): return (x << 768) % self.p
    def fr_to_mont(self, x): return (x << 384) % self.r
    def fr_from_mont(self, x): return x * pow(1 << 384, -1, self.r) % self.r
    def limbs(self, x, n): return [(x >> (64 * i)) & MASK64 for i in range(n)]
    def to_limbs(self, pt):
        if pt is None:
            return [0] * 24
        return self.limbs(self.fp_to_mont(pt[0]), 12) + self.limbs(self.fp_to_mont(pt[1]), 12)
    def fr_to_limbs(self, x): return self.limbs(self.fr_to_mont(x), 6)
    def fr_from_limbs(self, ls): return self.fr_from_mont(b.from_limbs(ls))

    def on_curve(self, group, pt): return b.ec_is_on_curve(self.fp, self.b[group], pt)
    def add(self, a, c): return b.ec_add(self.fp, a, c)
    def neg(self, a): return b.ec_neg(self.fp, a)
    def mul(self, a, k): return b.ec_mul(self.fp, a, k)

    def some_point(self, group, start=2):
        """Smallest x >= start with x^3 + b a square (p = 3 mod 4: sqrt = a^((p+1)/4)); NOT cofactor-cleared: the
        group law and the MSM (integer scalars) do not need the r-torsion."""
        x = start
        while True:
            rhs = (x * x * x + self.b[group]) % self.p
            y = pow(rhs, (self.p + 1) // 4, self.p)
            if y * y % self.p == rhs:
                return (x, min(y, self.p - y))
            x += 1

    def msm(self, pts, ks):
        """Integer-scalar MSM (no reduction mod r: the points need not lie in the r-torsion)."""
        if len(pts) != len(ks):
            raise ValueError("len(points) != len(scalars)")
        acc = None
        for pt, k in zip(pts, ks):
            acc = b.ec_add(self.fp, acc, b.ec_mul(self.fp, pt, k))
        return acc

    def fr_at(self, seed, i):
        """i-th canonical scalar of SplitMix64 stream `seed`: the 384-bit value of six stream words, mod r."""
        v = 0
        for k in range(6):
            v |= b.splitmix_at(seed, 6 * i + k) << (64 * k)
        return v % self.r

    def witness_like_scalar(self, seed, i):
        sel = b.splitmix_at(seed ^ 0x5EED5EED5EED5EED, i) % 100
        if sel < 40: return 0
        if sel < 65: return 1
        if sel < 90: return b.splitmix_at(seed, 6 * i)
        return self.fr_at(seed, i)

    # r-torsion generators for the library's SYNTHETIC Groth16 setup: [#E/r] * some_point(group), orders from the CM trace
    # (tests/test_groth16_curves.py re-derives them); the Suite-style aliases let the generic trapdoor oracle run on this curve
    g1_gen = (0xd82cbf66753123ed25942ffadbec116b901330673728468b1653febae12aa13a5d68dc240a36cfbe185365abc6cb0cc5042c14be9179f0c6c05fc952c93a806d5316c2b601db66bd557011eb2c7dd0c1891418e3ce0e512da946c2ca98c56f,
              0xa62fd67fdd91e327a96c02bc80385547a171b11241a2653b54d7359cd7569806b159fd05975390f644cd4d4d121918f1f84be0e364c557f196bd4095e732d987ca22009ba7577b80aaa35b641488679ed9ef0d43b32e776ad507137f20a2dd)
    g2_gen = (0x4cd0ed4bbb4bad28e9646093e4c6ab32a3a80f35265437deef8f50aa1221f459b249d724b2c155e2bf40a492ead210323e3f1c3e6991b9bcabe9da05882daf12d84f49c477fd322fc532b59d18f40b4cc45de6fbd67847acac591e8c5a93fa,
              0x5bbdc19380f7c707f6fe8680ef10cd46fa210a92bc4f56f1b92ab610ecb3fd508160dc51bab3ee5072aa3dedbe0766414556817439a0fbc33df16a239fc281edec5df53182dcf168c9171615a79353ac90858b4a2f2d20aeae7ededaead5b0)
    def g1_mul(self, a, k): return self.mul(a, k)
    def g2_mul(self, a, k): return self.mul(a, k)
    def g1_add(self, a, c): return self.add(a, c)
    def g2_add(self, a, c): return self.add(a, c)
    def g1_to_limbs(self, pt): return self.to_limbs(pt)
    def g2_to_limbs(self, pt): return self.to_limbs(pt)

    # scalar-field NTT (gnark-crypto ecc/bw6-761/fr/fft): GenericDomain only needs r, root, two_adicity, fr_gen
    two_adicity, fr_gen = 46, 15
    root = pow(15, (_P377 - 1) >> 46, _P377)

    def domain(self, m):
        return GenericDomain(self, m)

    def progression(self, base, seed, n):
        """P_i = [s0]B + i*[s1]B, (s0, s1) = the first two raw 256-bit blocks of stream `seed`
        (dvn_bw6761_points_synthetic)."""
        s0 = sum(b.splitmix_at(seed, k) << (64 * k) for k in range(4))
        s1 = sum(b.splitmix_at(seed, 4 + k) << (64 * k) for k in range(4))
        p0, d = self.mul(base, s0), self.mul(base, s1)
        out, cur = [], p0
        for _ in range(n):
            out.append(cur); cur = self.add(cur, d)
        return out


BW6_761 = Bw6761()


# --------------------------------------------------------------------------------------------------------
# Synthetic Groth16 instance with a known trapdoor on any suite (bn254_py.SyntheticR1CS / groth16_* restated over suite.r)
# --------------------------------------------------------------------------------------------------------
class SuiteRng(b.SplitMix64):
    def __init__(self, seed, r):
        super().__init__(seed); self.r = r
        self.words = (r.bit_length() + 63) // 64           # the width of the field's image: 4 or (BW6-761) 6 stream words
    def next_fr(self):
        v = 0
        for k in range(self.words):
            v |= self.next() << (64 * k)
        return v % self.r


class SyntheticR1CSOn:
    """n_c constraints (w[l1] + k w[l2]) * w[r] = w[o] over suite.r (dvn_<curve>_r1cs_synthetic)."""

    def __init__(self, suite: Suite, n_constraints: int, n_public: int, seed: int):
        R = suite.r
        rng = SuiteRng(seed, R)
        self.suite, self.n_public, self.n_constraints = suite, n_public, n_constraints
        w = [1] + [rng.next_fr() for _ in range(n_public - 1)]
        w += [rng.next_fr() for _ in range(4)]
        self.A, self.B, self.C = [], [], []
        for _ in range(n_constraints):
            m = len(w)
            l1 = rng.next() % m; l2 = rng.next() % m; k = rng.next() % 7 + 1; r_ = rng.next() % m
            w.append(((w[l1] + k * w[l2]) * w[r_]) % R)
            self.A.append([(l1, 1), (l2, k)] if l1 != l2 else [(l1, (1 + k) % R)])
            self.B.append([(r_, 1)]); self.C.append([(m, 1)])
        self.witness, self.n_wires = w, len(w)


def groth16_expected_from_trapdoor_on(cs: SyntheticR1CSOn, seed: int, r: int, s: int):
    """Closed-form proof (Ar, Bs, Krs) from the toxic waste of stream `seed` (tau, alpha, beta, gamma, delta)."""
    S = cs.suite; R = S.r
    rng = SuiteRng(seed, R)
    tau, alpha, beta, gamma, delta = (rng.next_fr() or 1 for _ in range(5))
    dom = S.domain(cs.n_constraints); n = dom.n
    zt = (pow(tau, n, R) - 1) % R
    # L_j(tau) = zt w^j / (n (tau - w^j)); the n denominators are inverted together (Montgomery's trick: one modular
    # inversion), which is what lets this checker run at the benchmarked sizes (n = 2^20) in seconds
    ws, den, wj = [], [], 1
    for _ in range(n):
        ws.append(wj); den.append(n * (tau - wj) % R); wj = wj * dom.gen % R
    assert all(den), "tau lies in the domain"
    pre, acc = [], 1
    for d in den:
        pre.append(acc); acc = acc * d % R
    inv = pow(acc, -1, R)
    L = [0] * n
    for j in range(n - 1, -1, -1):
        L[j] = zt * ws[j] % R * (inv * pre[j] % R) % R
        inv = inv * den[j] % R
    m = cs.n_wires
    At, Bt, Ct = [0] * m, [0] * m, [0] * m
    for rows, acc in ((cs.A, At), (cs.B, Bt), (cs.C, Ct)):
        for j, row in enumerate(rows):
            for i, c in row:
                acc[i] = (acc[i] + c * L[j]) % R
    dinv = pow(delta, -1, R)
    K = [((beta * At[i] + alpha * Bt[i] + Ct[i]) % R) * dinv % R for i in range(cs.n_public, m)]
    w = cs.witness
    a_tau = sum(wi * v for wi, v in zip(w, At)) % R
    b_tau = sum(wi * v for wi, v in zip(w, Bt)) % R
    c_tau = sum(wi * v for wi, v in zip(w, Ct)) % R
    h_tau = (a_tau * b_tau - c_tau) * pow(zt, -1, R) % R
    ar = (alpha + a_tau + r * delta) % R
    bs = (beta + b_tau + s * delta) % R
    priv = sum(w[cs.n_public + i] * k for i, k in enumerate(K)) % R
    krs = (priv + h_tau * zt % R * dinv + s * ar + r * bs - r * s % R * delta) % R
    return {"Ar": S.g1_mul(S.g1_gen, ar), "Bs": S.g2_mul(S.g2_gen, bs), "Krs": S.g1_mul(S.g1_gen, krs)}
