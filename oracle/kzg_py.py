"""TEST INFRASTRUCTURE ONLY -- CPU restatement (Python big ints) of the EIP-4844 / EIP-7594 blob KZG operations the
reference performs before every state-transition proof:

  Blob.ComputeCommitment          /root/reference/types/blobs.go:88-95      (gethkzg.BlobToCommitment)
  Blob.ComputeProof(z)            /root/reference/types/blobs.go:120-133    (gethkzg.ComputeProof -> proof, claim y)
  Blob.ComputeCellProofs          /root/reference/types/blobs.go:97-104     (gethkzg.ComputeCellProofs, 128 proofs)
  EvaluateBarycentricNative       /root/reference/crypto/blobs/barycentric.go:24-170  (IN TREE: restated line by line)
  Blob.ComputeBlobProof           /root/reference/types/blobs.go:108-117    (gethkzg.ComputeBlobProof, Fiat-Shamir point)

The three geth calls live in un-vendored modules (go-ethereum crypto/kzg4844 -> crate-crypto/go-eth-kzg; go.mod); what
is restated for them is the published consensus-spec algorithm (EIP-4844 polynomial-commitments.md: blob_to_kzg_commitment,
compute_kzg_proof_impl, compute_quotient_eval_within_domain; EIP-7594 polynomial-commitments-sampling.md:
compute_cells_and_kzg_proofs, coset_for_cell).  KZG proofs are unique group elements, so any correct algorithm yields
the same 48 bytes.

PINNED by reference data (tests/test_kzg.py, tests/golden/*): the trusted setup the reference embeds
(config/kzg_trusted_setup.txt: Lagrange and monomial G1 bases of the same tau -> sum L_i = [1], sum w^i L_i = [tau],
Lagrange-form commitment == monomial-form commitment), the 4096 bit-reversed roots of crypto/blobs/omega.go, the
2^32-th root of unity literal of barycentric.go:52 and the two test blobs of crypto/blobs/testdata/.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
N = 4096                       # field elements per blob
CELLS = 128                    # EIP-7594 CELLS_PER_EXT_BLOB
CELL = 64                      # FIELD_ELEMENTS_PER_CELL
ROOT_2_32 = 10238227357739495823651030575849232062558860180284477541189508159991286009131   # barycentric.go:52
G1_GEN = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)


def bitrev(i: int, bits: int) -> int:          # barycentric.go:191-200
    r = 0
    for k in range(bits):
        if (i >> k) & 1:
            r |= 1 << (bits - 1 - k)
    return r


def root_of_unity(order: int) -> int:          # barycentric.go:57-59 (order 4096 <-> exponent 2^20)
    log = order.bit_length() - 1
    assert order == 1 << log and log <= 32
    return pow(ROOT_2_32, 1 << (32 - log), R)


def domain_brp(order: int = N):                # barycentric.go:62-73: omega[i] = w^bitrev(i)
    w = root_of_unity(order)
    nat = [1] * order
    for i in range(1, order):
        nat[i] = nat[i - 1] * w % R
    bits = order.bit_length() - 1
    return [nat[bitrev(i, bits)] for i in range(order)]


# ---- G1 (y^2 = x^3 + 4 over Fp), Jacobian arithmetic; None = infinity --------------------------------------------------
def g1_on_curve(pt) -> bool:
    return pt is None or (pt[1] * pt[1] - pt[0] * pt[0] * pt[0] - 4) % P == 0


def jac_dbl(p):
    if p is None:
        return None
    X, Y, Z = p
    if Y == 0:
        return None
    A = X * X % P; B = Y * Y % P; C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P; F = E * E % P
    X3 = (F - 2 * D) % P
    return (X3, (E * (D - X3) - 8 * C) % P, 2 * Y * Z % P)


def jac_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    X1, Y1, Z1 = p; X2, Y2, Z2 = q
    Z1Z1 = Z1 * Z1 % P; Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P; U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P; S2 = Y2 * Z1 * Z1Z1 % P
    H = (U2 - U1) % P; Rr = (S2 - S1) % P
    if H == 0:
        return jac_dbl(p) if Rr == 0 else None
    HH = H * H % P; HHH = H * HH % P; V = U1 * HH % P
    X3 = (Rr * Rr - HHH - 2 * V) % P
    return (X3, (Rr * (V - X3) - S1 * HHH) % P, Z1 * Z2 * H % P)


def to_jac(pt):
    return None if pt is None else (pt[0], pt[1], 1)


def to_affine(p):
    if p is None:
        return None
    zi = pow(p[2], -1, P); z2 = zi * zi % P
    return (p[0] * z2 % P, p[1] * z2 * zi % P)


def g1_add(a, c):
    return to_affine(jac_add(to_jac(a), to_jac(c)))


def g1_mul(pt, k: int):
    acc, base = None, to_jac(pt)
    k %= R
    while k:
        if k & 1:
            acc = jac_add(acc, base)
        base = jac_dbl(base)
        k >>= 1
    return to_affine(acc)


def g1_msm(points, scalars, c: int = 8):
    """Textbook Pippenger (unsigned c-bit windows, running-sum bucket reduction, MSB-first combine)."""
    if len(points) != len(scalars):
        raise ValueError("len(points) != len(scalars)")
    jp = [to_jac(p) for p in points]
    ks = [k % R for k in scalars]
    nwin = (255 + c - 1) // c
    acc = None
    for w in range(nwin - 1, -1, -1):
        for _ in range(c):
            acc = jac_dbl(acc)
        buckets = [None] * (1 << c)
        for p, k in zip(jp, ks):
            d = (k >> (w * c)) & ((1 << c) - 1)
            if d and p is not None:
                buckets[d] = jac_add(buckets[d], p)
        run = tot = None
        for d in range((1 << c) - 1, 0, -1):
            run = jac_add(run, buckets[d])
            tot = jac_add(tot, run)
        acc = jac_add(acc, tot)
    return to_affine(acc)


# ---- ZCash compressed encoding of G1 (what KZGCommitment / KZGProof [48]byte hold, types/blobs.go:165,199) ------------
def g1_compress(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80
    if y > (P - 1) // 2:
        out[0] |= 0x20
    return bytes(out)


def g1_decompress(b48: bytes):
    assert len(b48) == 48 and b48[0] & 0x80, "not a compressed G1 encoding"
    if b48[0] & 0x40:
        assert b48[0] == 0xC0 and not any(b48[1:]), "bad infinity encoding"
        return None
    x = int.from_bytes(bytes([b48[0] & 0x1F]) + b48[1:], "big")
    assert x < P
    y2 = (x * x * x + 4) % P
    y = pow(y2, (P + 1) // 4, P)
    assert y * y % P == y2, "x is not on the curve"
    if (y > (P - 1) // 2) != bool(b48[0] & 0x20):
        y = P - y
    return (x, y)


class TrustedSetup:
    """config/kzg_trusted_setup.txt: '4096', '65', 4096 G1 Lagrange (natural order), 65 G2 monomial, 4096 G1 monomial."""

    def __init__(self, path: str):
        lines = open(path).read().split()
        n1, n2 = int(lines[0]), int(lines[1])
        assert (n1, n2) == (N, 65) and len(lines) == 2 + 2 * n1 + n2
        self.g1_lagrange_bytes = [bytes.fromhex(h) for h in lines[2:2 + n1]]
        self.g2_monomial_bytes = [bytes.fromhex(h) for h in lines[2 + n1:2 + n1 + n2]]
        self.g1_monomial_bytes = [bytes.fromhex(h) for h in lines[2 + n1 + n2:]]
        self.g1_lagrange = [g1_decompress(b) for b in self.g1_lagrange_bytes]
        self.g1_monomial = [g1_decompress(b) for b in self.g1_monomial_bytes]
        # the blob's i-th element is the evaluation at w^bitrev(i) (barycentric.go:70-73), so its basis point is L_bitrev(i)
        self.g1_lagrange_brp = [self.g1_lagrange[bitrev(i, 12)] for i in range(N)]


def blob_to_field_elements(blob: bytes):
    """32-byte big-endian field elements (barycentric.go:179-186); values >= r are rejected (EIP-4844 bytes_to_bls_field)."""
    if len(blob) != 32 * N:
        raise ValueError("blob length is %d, want 131072" % len(blob))
    out = [int.from_bytes(blob[32 * i:32 * i + 32], "big") for i in range(N)]
    if any(v >= R for v in out):
        raise ValueError("blob element is not a canonical field element")
    return out


def evaluate_barycentric(blob: bytes, z: int) -> int:
    """crypto/blobs/barycentric.go:24-170: y = (z^4096 - 1)/4096 * sum_i d_i w_i / (z - w_i), d_i at w_i = w^bitrev(i)."""
    d = blob_to_field_elements(blob)
    omega = domain_brp(N)
    z %= R
    for k, w in enumerate(omega):                       # :99-107 early exit
        if w == z:
            return d[k] % R
    s = 0
    for i in range(N):                                  # :118-150
        if d[i] == 0:
            continue
        s = (s + d[i] * omega[i] * pow((z - omega[i]) % R, -1, R)) % R
    factor = (pow(z, N, R) - 1) * pow(N, -1, R) % R     # :159-163
    return s * factor % R


def blob_to_commitment(ts: TrustedSetup, blob: bytes) -> bytes:
    return g1_compress(g1_msm(ts.g1_lagrange_brp, blob_to_field_elements(blob)))


def quotient_evaluations(blob: bytes, z: int):
    """EIP-4844 compute_kzg_proof_impl: evaluations of q(X) = (f(X) - y)/(X - z) on the bit-reversed domain, and y."""
    f = blob_to_field_elements(blob)
    omega = domain_brp(N)
    z %= R
    y = evaluate_barycentric(blob, z)
    q = [0] * N
    m = omega.index(z) if z in omega else -1
    for i in range(N):
        if i == m:
            continue
        q[i] = (f[i] - y) * pow((omega[i] - z) % R, -1, R) % R
    if m >= 0:                                          # compute_quotient_eval_within_domain
        acc = 0
        for i in range(N):
            if i == m:
                continue
            acc = (acc + (f[i] - y) * omega[i] * pow(z * ((z - omega[i]) % R) % R, -1, R)) % R
        q[m] = acc
    return q, y


def compute_proof(ts: TrustedSetup, blob: bytes, z: int):
    """-> (48-byte proof, y): what Blob.ComputeProof returns (types/blobs.go:120-133)."""
    q, y = quotient_evaluations(blob, z)
    return g1_compress(g1_msm(ts.g1_lagrange_brp, q)), y


def blob_coefficients(blob: bytes):
    """Monomial coefficients of the blob polynomial: f_j = (1/N) sum_i d_i w_i^-j over the bit-reversed domain."""
    d = blob_to_field_elements(blob)
    nat = [0] * N
    for i in range(N):
        nat[bitrev(i, 12)] = d[i]                        # evaluations in natural order
    # iterative radix-2 inverse transform (natural in, natural out)
    a = [nat[bitrev(i, 12)] for i in range(N)]
    winv = pow(root_of_unity(N), -1, R)
    size = 2
    while size <= N:
        step = pow(winv, N // size, R)
        for start in range(0, N, size):
            w = 1
            for j in range(size // 2):
                u, v = a[start + j], a[start + j + size // 2] * w % R
                a[start + j], a[start + j + size // 2] = (u + v) % R, (u - v) % R
                w = w * step % R
        size <<= 1
    ninv = pow(N, -1, R)
    return [x * ninv % R for x in a]


def cell_coset_shift(k: int) -> int:
    """EIP-7594 coset_for_cell: cell k holds the evaluations at roots_of_unity_brp(8192)[64k : 64k+64] = h_k * <w_64>,
    h_k = w_8192^bitrev_128(k)."""
    return pow(root_of_unity(2 * N), bitrev(k, 7), R)


def cell_quotient_coefficients(coeffs, k: int):
    """q_k(X) = (f(X) - I_k(X)) / (X^64 - h_k^64): synthetic division, the remainder I_k is dropped."""
    c = pow(cell_coset_shift(k), CELL, R)
    q = [0] * (N - CELL)
    for j in range(N - CELL - 1, -1, -1):
        hi = q[j + CELL] if j + CELL < N - CELL else 0
        q[j] = (coeffs[j + CELL] + c * hi) % R
    return q


def compute_cell_proof(ts: TrustedSetup, blob: bytes, k: int, coeffs=None) -> bytes:
    coeffs = coeffs if coeffs is not None else blob_coefficients(blob)
    return g1_compress(g1_msm(ts.g1_monomial[:N - CELL], cell_quotient_coefficients(coeffs, k)))


def cell_evaluations(coeffs, k: int):
    """The 64 evaluations cell k of the extended blob holds (Horner; used to check a cell proof's remainder)."""
    h = cell_coset_shift(k)
    w64 = root_of_unity(CELL)
    out = []
    for j in range(CELL):
        x = h * pow(w64, bitrev(j, 6), R) % R
        acc = 0
        for cf in reversed(coeffs):
            acc = (acc * x + cf) % R
        out.append(acc)
    return out


def compute_challenge(blob: bytes, commitment: bytes) -> int:
    """EIP-4844 compute_challenge: sha256("FSBLOBVERIFY_V1_" || uint128_be(4096) || blob || commitment) mod r."""
    import hashlib
    data = b"FSBLOBVERIFY_V1_" + N.to_bytes(16, "big") + blob + commitment
    return int.from_bytes(hashlib.sha256(data).digest(), "big") % R


def compute_blob_proof(ts: TrustedSetup, blob: bytes, commitment: bytes) -> bytes:
    """Blob.ComputeBlobProof (types/blobs.go:108-117, gethkzg.ComputeBlobProof): EIP-4844 compute_blob_kzg_proof."""
    return compute_proof(ts, blob, compute_challenge(blob, commitment))[0]
