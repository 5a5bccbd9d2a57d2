"""TEST INFRASTRUCTURE (oracle): writer and reader of gnark's binary Groth16 proving-key format for BN254.

The reference writes its proving keys with `pk.WriteTo(f)` (/root/reference/cmd/circuit-compile/main.go:507-513) and reads
them back with `pk.UnsafeReadFrom(f)` (/root/reference/circuits/artifacts.go:391-406).  Both live in the un-vendored module
github.com/consensys/gnark v0.14.1-0.20260126121332-407111efab55 (backend/groth16/bn254/marshal.go) over gnark-crypto
v0.19.3-0.20260126145145-b5cf053fbc34 (ecc/bn254/marshal.go, ecc/bn254/fr/fft/domain.go, ecc/bn254/fr/pedersen): there is no
key file and no Go toolchain in reach, so this module restates the PUBLISHED layout and the device-side reader
(csrc/dvn_artifacts.hip) is checked against it -- PARITY UNPINNED by reference data.

Layout (all integers big-endian):
  fft.Domain.WriteTo   uint64 Cardinality, then CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen,
                       FrMultiplicativeGenInv as 32-byte canonical field elements, then the withPrecompute bool (1 byte)
  &G1.Alpha, &G1.Beta, &G1.Delta, G1.A, G1.B, G1.Z, G1.K, &G2.Beta, &G2.Delta, G2.B   (slices: uint32 length + points)
  uint64 nbWires, uint64 NbInfinityA, uint64 NbInfinityB, InfinityA, InfinityB ([]bool, one byte each)
  uint32 len(CommitmentKeys), then per key pedersen.ProvingKey.WriteTo: Basis, BasisExpSigma (slices)
Points (ecc/bn254/marshal.go): flag = top two bits of the first byte: 00 uncompressed (G1 X|Y; G2 X.A1|X.A0|Y.A1|Y.A0;
infinity all zero), 10 / 11 compressed with the lexicographically smallest / largest y (G1: X; G2: X.A1|X.A0), 01 compressed
infinity.  y is "largest" when y > (p-1)/2; for Fp2 by A1 unless A1 = 0, then by A0.
"""
from __future__ import annotations
import struct

from . import bn254_py as o

P, R = o.P, o.R
M_UNCOMPRESSED, M_INFINITY, M_SMALLEST, M_LARGEST = 0b00 << 6, 0b01 << 6, 0b10 << 6, 0b11 << 6


def _lex_largest(y: int) -> bool:
    return y > (P - 1) // 2


def _lex_largest2(y) -> bool:
    return _lex_largest(y[1]) if y[1] % P else _lex_largest(y[0])


def g1_bytes(pt, raw: bool) -> bytes:
    if raw:
        return bytes(64) if pt is None else pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")
    if pt is None:
        return bytes([M_INFINITY]) + bytes(31)
    b = bytearray(pt[0].to_bytes(32, "big"))
    b[0] |= M_LARGEST if _lex_largest(pt[1]) else M_SMALLEST
    return bytes(b)


def g2_bytes(pt, raw: bool) -> bytes:
    if raw:
        if pt is None:
            return bytes(128)
        (x0, x1), (y0, y1) = pt
        return b"".join(v.to_bytes(32, "big") for v in (x1, x0, y1, y0))
    if pt is None:
        return bytes([M_INFINITY]) + bytes(63)
    (x0, x1), y = pt
    b = bytearray(x1.to_bytes(32, "big") + x0.to_bytes(32, "big"))
    b[0] |= M_LARGEST if _lex_largest2(y) else M_SMALLEST
    return bytes(b)


def _fp_sqrt(a: int):
    y = pow(a, (P + 1) // 4, P)
    return y if y * y % P == a % P else None


def g1_from_bytes(b: bytes):
    """-> (point or None, bytes consumed)"""
    flag = b[0] & 0xC0
    if flag == M_UNCOMPRESSED:
        x, y = int.from_bytes(b[:32], "big"), int.from_bytes(b[32:64], "big")
        return (None if x == 0 and y == 0 else (x, y)), 64
    if flag == M_INFINITY:
        return None, 32
    x = int.from_bytes(bytes([b[0] & 0x3F]) + b[1:32], "big")
    y = _fp_sqrt((x * x * x + 3) % P)
    assert y is not None, "not on the curve"
    if _lex_largest(y) != (flag == M_LARGEST):
        y = P - y
    return (x, y), 32


def _slice(points, enc) -> bytes:
    return struct.pack(">I", len(points)) + b"".join(enc(p) for p in points)


def write_proving_key(log_n: int, n_wires: int, A, B, Z, K, B2, idxA, idxB, consts, commitment_keys=(), raw: bool = False,
                      domain_flag: bool = True, bitmap_prefix: bool = False) -> bytes:
    """A, B, Z, K: lists of affine G1 points (int pairs or None); B2: affine G2 points ((x0, x1), (y0, y1));
    idxA / idxB: the wires whose A / B points are kept (all others are flagged in InfinityA / InfinityB);
    consts = (alpha1, beta1, delta1, beta2, delta2); commitment_keys: [(Basis, BasisExpSigma), ...] lists of G1 points."""
    n = 1 << log_n
    dom = o.Domain(n)
    g = 5
    fr = lambda v: (v % R).to_bytes(32, "big")
    out = [struct.pack(">Q", n), fr(pow(n, -1, R)), fr(dom.gen), fr(pow(dom.gen, -1, R)), fr(g), fr(pow(g, -1, R))]
    if domain_flag:
        out.append(b"\x01")                                   # withPrecompute
    e1 = lambda p: g1_bytes(p, raw)
    e2 = lambda p: g2_bytes(p, raw)
    alpha1, beta1, delta1, beta2, delta2 = consts
    out += [e1(alpha1), e1(beta1), e1(delta1), _slice(A, e1), _slice(B, e1), _slice(Z, e1), _slice(K, e1), e2(beta2), e2(delta2), _slice(B2, e2)]
    keepA, keepB = set(int(i) for i in idxA), set(int(i) for i in idxB)
    assert len(keepA) == len(A) and len(keepB) == len(B) == len(B2)
    infA = bytes(0 if i in keepA else 1 for i in range(n_wires))
    infB = bytes(0 if i in keepB else 1 for i in range(n_wires))
    pre = struct.pack(">I", n_wires) if bitmap_prefix else b""       # the variant in which []bool carries its uint32 length
    out += [struct.pack(">QQQ", n_wires, n_wires - len(A), n_wires - len(B)), pre, infA, pre, infB, struct.pack(">I", len(commitment_keys))]
    for basis, basis_sigma in commitment_keys:
        out += [_slice(basis, e1), _slice(basis_sigma, e1)]
    return b"".join(out)


def key_arrays_to_points(arrays: dict):
    """The uint64 Montgomery images of Groth16Key.export_arrays -> the point lists write_proving_key takes."""
    g1 = lambda rows: [o.g1_from_limbs([int(v) for v in r]) for r in rows]
    g2 = lambda rows: [o.g2_from_limbs([int(v) for v in r]) for r in rows]
    c = [int(v) for v in arrays["consts"]]
    consts = (o.g1_from_limbs(c[0:8]), o.g1_from_limbs(c[8:16]), o.g1_from_limbs(c[16:24]), o.g2_from_limbs(c[24:40]), o.g2_from_limbs(c[40:56]))
    return g1(arrays["A"]), g1(arrays["B"]), g1(arrays["Z"]), g1(arrays["K"]), g2(arrays["B2"]), consts


# ------------------------------------------------------------------------------------------------------------------------
# The same file layout on the curves whose base field leaves three spare bits in the first byte (gnark-crypto
# ecc/bls12-377/marshal.go, ecc/bw6-761/marshal.go): the voteverifier's and the aggregator's proving keys
# (/root/reference/sequencer/circuit_artifacts.go:39-76).  Flag = top THREE bits of the first byte: 000 uncompressed,
# 010 uncompressed infinity, 100 / 101 compressed with the smallest / largest y, 110 compressed infinity.  Coordinates are
# big-endian canonical integers of the base field's byte width (48 / 96); scalar-field elements of the Domain header of theirs
# (32 / 48).  PARITY UNPINNED, as above.
# ------------------------------------------------------------------------------------------------------------------------
M3_UNCOMPRESSED, M3_UNCOMPRESSED_INF, M3_SMALLEST, M3_LARGEST, M3_COMPRESSED_INF = 0b000 << 5, 0b010 << 5, 0b100 << 5, 0b101 << 5, 0b110 << 5


class SuiteKeyFormat:
    """Writer for a curves_py suite: `quadratic` = G2 lives over Fp2 (BLS12-377) or over Fp (BW6-761)."""

    def __init__(self, suite, quadratic: bool):
        self.S, self.quadratic = suite, quadratic
        self.p, self.r = suite.p, suite.r
        self.cb = (self.p.bit_length() + 63) // 64 * 8
        self.frb = (self.r.bit_length() + 63) // 64 * 8
        assert self.cb * 8 - self.p.bit_length() >= 3

    def lex_largest(self, y: int) -> bool:
        return y > (self.p - 1) // 2

    def lex_largest2(self, y) -> bool:
        return self.lex_largest(y[1]) if y[1] % self.p else self.lex_largest(y[0])

    def _coords(self, vals) -> bytearray:
        return bytearray(b"".join(int(v).to_bytes(self.cb, "big") for v in vals))

    def fp_point_bytes(self, pt, raw: bool) -> bytes:
        """A point of a curve over Fp (G1 of both curves, G2 of BW6-761)."""
        if raw:
            if pt is None:
                return bytes([M3_UNCOMPRESSED_INF]) + bytes(2 * self.cb - 1)
            return bytes(self._coords(pt))
        if pt is None:
            return bytes([M3_COMPRESSED_INF]) + bytes(self.cb - 1)
        b = self._coords([pt[0]])
        b[0] |= M3_LARGEST if self.lex_largest(pt[1]) else M3_SMALLEST
        return bytes(b)

    def fp2_point_bytes(self, pt, raw: bool) -> bytes:
        """A point of a curve over Fp2: X.A1 | X.A0 (| Y.A1 | Y.A0)."""
        if raw:
            if pt is None:
                return bytes([M3_UNCOMPRESSED_INF]) + bytes(4 * self.cb - 1)
            (x0, x1), (y0, y1) = pt
            return bytes(self._coords([x1, x0, y1, y0]))
        if pt is None:
            return bytes([M3_COMPRESSED_INF]) + bytes(2 * self.cb - 1)
        (x0, x1), y = pt
        b = self._coords([x1, x0])
        b[0] |= M3_LARGEST if self.lex_largest2(y) else M3_SMALLEST
        return bytes(b)

    def g1_bytes(self, pt, raw): return self.fp_point_bytes(pt, raw)
    def g2_bytes(self, pt, raw): return self.fp2_point_bytes(pt, raw) if self.quadratic else self.fp_point_bytes(pt, raw)

    def write_proving_key(self, log_n: int, n_wires: int, A, B, Z, K, B2, idxA, idxB, consts, commitment_keys=(), raw: bool = False,
                          domain_flag: bool = True, bitmap_prefix: bool = False) -> bytes:
        n = 1 << log_n
        dom = self.S.domain(n)
        g = dom.coset
        fr = lambda v: (v % self.r).to_bytes(self.frb, "big")
        out = [struct.pack(">Q", n), fr(pow(n, -1, self.r)), fr(dom.gen), fr(pow(dom.gen, -1, self.r)), fr(g), fr(pow(g, -1, self.r))]
        if domain_flag:
            out.append(b"\x01")
        e1 = lambda p: self.g1_bytes(p, raw)
        e2 = lambda p: self.g2_bytes(p, raw)
        alpha1, beta1, delta1, beta2, delta2 = consts
        out += [e1(alpha1), e1(beta1), e1(delta1), _slice(A, e1), _slice(B, e1), _slice(Z, e1), _slice(K, e1), e2(beta2), e2(delta2), _slice(B2, e2)]
        keepA, keepB = set(int(i) for i in idxA), set(int(i) for i in idxB)
        assert len(keepA) == len(A) and len(keepB) == len(B) == len(B2)
        infA = bytes(0 if i in keepA else 1 for i in range(n_wires))
        infB = bytes(0 if i in keepB else 1 for i in range(n_wires))
        pre = struct.pack(">I", n_wires) if bitmap_prefix else b""
        out += [struct.pack(">QQQ", n_wires, n_wires - len(A), n_wires - len(B)), pre, infA, pre, infB, struct.pack(">I", len(commitment_keys))]
        for basis, basis_sigma in commitment_keys:
            out += [_slice(basis, e1), _slice(basis_sigma, e1)]
        return b"".join(out)

    def key_arrays_to_points(self, arrays: dict):
        """uint64 Montgomery images of Groth16Key.export_arrays -> the point lists write_proving_key takes."""
        S, L = self.S, self.cb // 8
        rinv = pow(1 << (64 * L), -1, self.p)
        val = lambda ws: sum(int(v) << (64 * i) for i, v in enumerate(ws)) * rinv % self.p
        def fp_pt(row):
            x, y = val(row[0:L]), val(row[L:2 * L])
            return None if not any(int(v) for v in row) else (x, y)
        def fp2_pt(row):
            if not any(int(v) for v in row):
                return None
            return ((val(row[0:L]), val(row[L:2 * L])), (val(row[2 * L:3 * L]), val(row[3 * L:4 * L])))
        g2 = fp2_pt if self.quadratic else fp_pt
        c = [int(v) for v in arrays["consts"]]
        a1, a2 = 2 * L, (4 * L if self.quadratic else 2 * L)
        consts = (fp_pt(c[0:a1]), fp_pt(c[a1:2 * a1]), fp_pt(c[2 * a1:3 * a1]), g2(c[3 * a1:3 * a1 + a2]), g2(c[3 * a1 + a2:3 * a1 + 2 * a2]))
        return ([fp_pt(r) for r in arrays["A"]], [fp_pt(r) for r in arrays["B"]], [fp_pt(r) for r in arrays["Z"]], [fp_pt(r) for r in arrays["K"]],
                [g2(r) for r in arrays["B2"]], consts)


def suite_format(name: str) -> SuiteKeyFormat:
    from . import curves_py as cp
    return {"bls12377": SuiteKeyFormat(cp.BLS12_377, True), "bw6761": SuiteKeyFormat(cp.BW6_761, False)}[name]
