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
                      domain_flag: bool = True) -> bytes:
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
    out += [struct.pack(">QQQ", n_wires, n_wires - len(A), n_wires - len(B)), infA, infB, struct.pack(">I", len(commitment_keys))]
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
