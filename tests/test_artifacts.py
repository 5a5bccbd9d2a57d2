"""gnark proving-key artifact reader: the binary layout `pk.WriteTo` produces
(/root/reference/cmd/circuit-compile/main.go:507-513) and `pk.UnsafeReadFrom` consumes
(/root/reference/circuits/artifacts.go:391-406), read by csrc/dvn_artifacts.hip into a device-resident key.

PARITY UNPINNED: the format lives in the un-vendored gnark module and no key file is reachable, so the reader is exercised
against this repository's own writer (oracle/gnark_key_format_py.py), which restates the published layout.  What IS checked
end to end: a key exported from the Setup mirror, written to a file in that layout (compressed and raw), loaded, and proved
from reproduces the original key's proof bit for bit and the trapdoor closed form.
"""
import hashlib

import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def mont(o, x):
    return np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)


def _tiny_image(opy, fmt, raw, domain_flag=True, commitment=False, n_wires=9):
    o = opy
    g = lambda k: o.g1_mul(o.G1_GEN, k)
    h = lambda k: o.g2_mul(o.G2_GEN, k)
    idxA, idxB = [0, 2, 3, 5], [1, 2, 8]
    A = [g(3), g(4), None, g(6)]                     # a kept point may still be the point at infinity
    B = [g(7), g(8), g(9)]
    B2 = [h(7), h(8), h(9)]
    Z = [g(20 + i) for i in range(4)]
    K = [g(30 + i) for i in range(n_wires - 2 - (2 if commitment else 0))]
    ck = [([g(50), g(51)], [g(60), g(61)])] if commitment else []
    return fmt.write_proving_key(2, n_wires, A, B, Z, K, B2, idxA, idxB, (g(11), g(12), g(13), h(12), h(13)), ck, raw=raw, domain_flag=domain_flag)


@pytest.mark.parametrize("raw", [False, True])
@pytest.mark.parametrize("domain_flag", [True, False])
def test_scan_reports_the_sections(pkg, opy, raw, domain_flag):
    """Host only: section walk of a written image, compressed / raw, with and without the domain's flag byte."""
    from oracle import gnark_key_format_py as fmt
    img = _tiny_image(opy, fmt, raw, domain_flag, commitment=True)
    info = pkg.gnark_pk_scan(img)
    assert info == {"log_n": 2, "raw": int(raw), "n_wires": 9, "A": 4, "B": 3, "Z": 4, "K": 5, "B2": 3, "n_commitments": 1, "basis0": 2,
                    "domain_flag": int(domain_flag), "bitmap_prefix": 0}
    for bad in (img[:-1], img + b"\x00", img[:200], b"\x00" * 64, img[:8] + b"\xff" * (len(img) - 8)):
        with pytest.raises(pkg.DvnError):
            pkg.gnark_pk_scan(bad)
    # a count that disagrees with the bitmaps is refused
    broken = bytearray(img)
    off = img.rindex(b"\x00\x00\x00\x00\x00\x00\x00\x09")         # nbWires (the last 8-byte big-endian 9 in the image)
    broken[off + 15] ^= 1                                         # NbInfinityA
    with pytest.raises(pkg.DvnError):
        pkg.gnark_pk_scan(bytes(broken))


def test_writer_point_encodings_round_trip(opy):
    """The oracle's own encodings: flags, lexicographic rule, infinity, raw form."""
    from oracle import gnark_key_format_py as fmt
    o = opy
    for k in (1, 2, 5, 12345, R - 1):
        pt = o.g1_mul(o.G1_GEN, k)
        for raw in (False, True):
            b = fmt.g1_bytes(pt, raw)
            assert len(b) == (64 if raw else 32) and fmt.g1_from_bytes(b) == (pt, len(b))
        neg = (pt[0], o.P - pt[1])
        assert (fmt.g1_bytes(pt, False)[0] ^ fmt.g1_bytes(neg, False)[0]) & 0xC0 == 0x40      # smallest 10 vs largest 11
    assert fmt.g1_bytes(None, False)[0] == 0x40 and fmt.g1_from_bytes(fmt.g1_bytes(None, False)) == (None, 32)
    assert fmt.g1_from_bytes(bytes(64)) == (None, 64)
    q = o.g2_mul(o.G2_GEN, 77)
    b = fmt.g2_bytes(q, False)
    assert len(b) == 64 and b[0] & 0x80 and int.from_bytes(bytes([b[0] & 0x3F]) + b[1:32], "big") == q[0][1]   # X.A1 first


@pytest.mark.gpu
@pytest.mark.parametrize("n_c,raw,domain_flag", [(1500, False, True), (1500, True, True), (1500, False, False), (1 << 16, False, True)])
def test_key_file_round_trip_proves(ctx, pkg, opy, orc, tmp_path, n_c, raw, domain_flag):
    """Setup -> export -> WRITE the gnark layout -> dvn_groth16_pk_load_file -> prove from the solver's output: the proof of the
    original key, and the trapdoor closed form.  Also: content hash check, cache hit on the second load.  At 2^16 constraints
    every point slice holds >= 2^15 points, so the loader builds expanded tables (the branch every real key takes) and the
    proof runs over them."""
    from oracle import gnark_key_format_py as fmt
    o = opy
    cs = pkg.R1CS(7, n_c, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    arrays = key.export_arrays(cs.n_wires, cs.n_public)
    A, B, Z, K, B2, consts = fmt.key_arrays_to_points(arrays)
    img = fmt.write_proving_key(key.log_n, cs.n_wires, A, B, Z, K, B2, arrays["idxA"], arrays["idxB"], consts, raw=raw, domain_flag=domain_flag)
    digest = hashlib.sha256(img).hexdigest()
    path = tmp_path / digest                                   # the node names artifacts by their hash
    path.write_bytes(img)
    with pytest.raises(pkg.DvnError) as e:
        pkg.LoadedKey(ctx, path, cs.n_public, sha256_hex="00" * 32)
    assert "hash mismatch" in str(e.value)
    lk = pkg.LoadedKey(ctx, path, cs.n_public, sha256_hex=digest)
    assert lk.log_n == key.log_n
    if n_c >= 1 << 16:
        assert min(lk.sizes[k] for k in ("A", "B", "K", "Z", "B2")) >= 1 << 15, "every slice must take the expanded-table branch"
    ref = o.SyntheticR1CS(n_c, 9, 7)
    a, b, c = (np.array([o.limbs(o.to_mont(v, R)) for v in col], dtype=np.uint64) for col in ref.abc())
    r, s = mont(o, 1234567), mont(o, 7654321)
    got = lk.prove_solved(cs.witness(), a, b, c, r, s)
    want = key.prove(cs.witness(), r, s)
    _, closed = orc.groth16_synthetic_expected(7, n_c, 9, 3, r, s)
    for k in ("Ar", "Bs", "Krs"):
        assert (got[k] == want[k]).all() and (got[k] == closed[k]).all(), k
    again = pkg.LoadedKey(ctx, path, cs.n_public, sha256_hex=digest.upper())
    assert again.h.value == lk.h.value, "the second load of the same content returns the resident key"
    lk.free()                                                  # cached keys outlive their handles ...
    got2 = again.prove_solved(cs.witness(), a, b, c, r, s)     # ... and keep proving
    assert (got2["Krs"] == closed["Krs"]).all()
    # wrong n_public: the K count no longer matches
    with pytest.raises(pkg.DvnError):
        pkg.LoadedKey(ctx, img, cs.n_public + 1)
    key.free(); cs.free()


@pytest.mark.gpu
def test_key_image_with_commitment_key(ctx, pkg, opy, orc):
    """A key whose file carries one commitment key: K leaves the committed wires out (the proof's Krs lacks exactly their
    terms), and the loaded Basis / BasisExpSigma feed dvn_pedersen_commit."""
    from oracle import gnark_key_format_py as fmt
    o = opy
    n_c = 600
    cs = pkg.R1CS(7, n_c, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    arrays = key.export_arrays(cs.n_wires, cs.n_public)
    A, B, Z, K, B2, consts = fmt.key_arrays_to_points(arrays)
    committed = [cs.n_public + 1, cs.n_public + 5, cs.n_public + 6, cs.n_wires - 1]
    keepK = [i for i in range(cs.n_public, cs.n_wires) if i not in committed]
    Kc = [K[i - cs.n_public] for i in keepK]
    basis_l = orc.g1_synthetic_points(81, len(committed))
    sigma = 0x1234567890ABCDEF
    sig_l = np.array([orc.g1_mul(basis_l[i], sigma) for i in range(len(committed))], dtype=np.uint64)
    ck = [([o.g1_from_limbs([int(v) for v in r]) for r in basis_l], [o.g1_from_limbs([int(v) for v in r]) for r in sig_l])]
    img = fmt.write_proving_key(key.log_n, cs.n_wires, A, B, Z, Kc, B2, arrays["idxA"], arrays["idxB"], consts, ck)
    assert pkg.gnark_pk_scan(img)["basis0"] == len(committed)
    lk = pkg.LoadedKey(ctx, img, cs.n_public, committed_wires=committed)
    w = cs.witness()
    ref = o.SyntheticR1CS(n_c, 9, 7)
    a, b, c = (np.array([o.limbs(o.to_mont(v, R)) for v in col], dtype=np.uint64) for col in ref.abc())
    r, s = mont(o, 5), mont(o, 6)
    got = lk.prove_solved(w, a, b, c, r, s)
    _, closed = orc.groth16_synthetic_expected(7, n_c, 9, 3, r, s)
    assert (got["Ar"] == closed["Ar"]).all() and (got["Bs"] == closed["Bs"]).all()
    missing = orc.msm_g1(arrays["K"][[i - cs.n_public for i in committed]], w[committed])      # sum of the committed wires' K terms
    assert (orc.g1_add(got["Krs"], missing) == closed["Krs"]).all()
    bs, bsg = lk.commitment_bases()
    cm, pok = ctx.pedersen_commit(bs, bsg, w[committed])
    assert (cm == orc.msm_g1(basis_l, w[committed])).all() and (pok == orc.g1_mul(cm, sigma)).all()
    st = lk.last_stats()
    assert st["log_n"] == key.log_n and st["msm"]["G1.K"]["pairs"] == len(keepK) and st["bytes"]["host_to_device"] == w.nbytes + 3 * a.nbytes
    lk.free(); key.free(); cs.free()


@pytest.mark.gpu
def test_malformed_points_are_rejected(ctx, pkg, opy):
    from oracle import gnark_key_format_py as fmt
    img = bytearray(_tiny_image(opy, fmt, raw=False))
    off = 8 + 5 * 32 + 1 + 3 * 32 + 4                             # first point of G1.A
    bad = bytearray(img); bad[off + 31] ^= 1                      # another x: on the curve only by chance; try a few
    seen = False
    for delta in range(1, 8):
        bad = bytearray(img); bad[off + 31] = (img[off + 31] + delta) & 0xFF
        try:
            k = pkg.LoadedKey(ctx, bytes(bad), 2)
            k.free()
        except pkg.DvnError as e:
            assert "not on the curve" in str(e)
            seen = True
            break
    assert seen
    mixed = bytearray(img); mixed[off] &= 0x3F                    # a raw-flagged point inside a compressed array
    with pytest.raises(pkg.DvnError):
        pkg.LoadedKey(ctx, bytes(mixed), 2)
    ok = pkg.LoadedKey(ctx, bytes(img), 2)
    ok.free()


def test_scan_survives_mutated_images(pkg, opy):
    """A key file is input from disk: the host-side section walk must reject damaged images with an error, never read out of
    bounds.  Truncations at every section border +-1, single-byte and 32-bit-count mutations, random tails."""
    import random
    from oracle import gnark_key_format_py as fmt
    rng = random.Random(11)
    for raw in (False, True):
        img = _tiny_image(opy, fmt, raw, True, commitment=True)
        good = pkg.gnark_pk_scan(img)
        n_ok = n_err = 0
        cuts = sorted(set(list(range(0, len(img), 37)) + [len(img) - k for k in range(1, 70)]))
        variants = [img[:c] for c in cuts]
        for _ in range(600):
            b = bytearray(img)
            k = rng.randrange(4)
            if k == 0:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            elif k == 1:
                off = rng.randrange(len(b) - 4); b[off:off + 4] = rng.getrandbits(32).to_bytes(4, "big")
            elif k == 2:
                off = rng.randrange(len(b) - 8); b[off:off + 8] = (rng.getrandbits(rng.choice((8, 20, 40, 63)))).to_bytes(8, "big")
            else:
                b += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 40)))
            variants.append(bytes(b))
        for v in variants:
            try:
                info = pkg.gnark_pk_scan(v)
                n_ok += 1
                assert info["n_wires"] == good["n_wires"] and info["A"] <= good["n_wires"]        # a mutation inside point bytes
            except pkg.DvnError:
                n_err += 1
        assert n_err > 100 and n_ok > 0, (n_ok, n_err)


@pytest.mark.gpu
def test_concurrent_loads_of_one_file_share_one_key(ctx, pkg, opy, tmp_path):
    """cgo calls arrive on arbitrary threads: several goroutines asking for the same artifact at once end up with ONE resident
    key (the losers of the race drop their copy)."""
    import threading
    from oracle import gnark_key_format_py as fmt
    img = _tiny_image(opy, fmt, raw=False)
    digest = hashlib.sha256(img).hexdigest()
    path = tmp_path / digest
    path.write_bytes(img)
    got, errs = [], []

    def load():
        try:
            got.append(pkg.LoadedKey(ctx, path, 2, sha256_hex=digest).h.value)
        except Exception as e:                                     # noqa: BLE001 - reported below
            errs.append(e)
    th = [threading.Thread(target=load) for _ in range(6)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs and len(got) == 6 and len(set(got)) == 1


# ---------------------------------------------------------------------------------------------------------------------
# The other two keys a node loads at boot (/root/reference/sequencer/circuit_artifacts.go:39-76): voteverifier (BLS12-377) and
# aggregator (BW6-761).  Same layout, 48- / 96-byte coordinates, three flag bits.  PARITY UNPINNED like the BN254 reader.
# ---------------------------------------------------------------------------------------------------------------------
def _suite(curve):
    from oracle import curves_py as cp
    return {"bls12377": cp.BLS12_377, "bw6761": cp.BW6_761}[curve]


def _tiny_image_on(curve, raw, domain_flag=True, bitmap_prefix=False, commitment=True):
    from oracle import gnark_key_format_py as fmt
    F, S = fmt.suite_format(curve), _suite(curve)
    g = lambda k: S.g1_mul(S.g1_gen, k)
    h = lambda k: S.g2_mul(S.g2_gen, k)
    idxA, idxB = [0, 2, 3, 5], [1, 2, 8]
    A = [g(3), g(4), None, g(6)]
    B = [g(7), g(8), g(9)]
    B2 = [h(7), h(8), h(9)]
    Z = [g(20 + i) for i in range(4)]
    K = [g(30 + i) for i in range(9 - 2 - (2 if commitment else 0))]
    ck = [([g(50), g(51)], [g(60), g(61)])] if commitment else []
    return F.write_proving_key(2, 9, A, B, Z, K, B2, idxA, idxB, (g(11), g(12), g(13), h(12), h(13)), ck, raw=raw, domain_flag=domain_flag,
                               bitmap_prefix=bitmap_prefix)


@pytest.mark.parametrize("curve", ["bls12377", "bw6761"])
@pytest.mark.parametrize("raw", [False, True])
def test_scan_other_curves_and_bitmap_variants(pkg, curve, raw):
    """Host only: section walk on the voteverifier / aggregator curves, both []bool variants reported, wrong-curve images and
    inconsistent Domain headers refused."""
    for domain_flag in (True, False):
        for prefix in (False, True):
            img = _tiny_image_on(curve, raw, domain_flag, prefix)
            info = pkg.gnark_pk_scan(img, curve)
            assert info == {"log_n": 2, "raw": int(raw), "n_wires": 9, "A": 4, "B": 3, "Z": 4, "K": 5, "B2": 3, "n_commitments": 1, "basis0": 2,
                            "domain_flag": int(domain_flag), "bitmap_prefix": int(prefix)}
    img = _tiny_image_on(curve, raw)
    other = "bw6761" if curve == "bls12377" else "bls12377"
    for c in ("bn254", other):
        with pytest.raises(pkg.DvnError):
            pkg.gnark_pk_scan(img, c)
    for bad in (img[:-1], img + b"\x00", img[:300]):
        with pytest.raises(pkg.DvnError):
            pkg.gnark_pk_scan(bad, curve)
    # a Domain header whose fields do not agree (Generator replaced by its square: no longer of order n) is named as such
    frb = 32 if curve == "bls12377" else 48
    S = _suite(curve)
    gen = int.from_bytes(img[8 + frb:8 + 2 * frb], "big")
    broken = img[:8 + frb] + (gen * gen % S.r).to_bytes(frb, "big") + img[8 + 2 * frb:]
    with pytest.raises(pkg.DvnError) as e:
        pkg.gnark_pk_scan(broken, curve)
    assert "primitive root" in str(e.value)


def test_scan_bn254_bitmap_prefix_variant_and_domain_check(pkg, opy):
    from oracle import gnark_key_format_py as fmt
    o = opy
    g = lambda k: o.g1_mul(o.G1_GEN, k)
    h = lambda k: o.g2_mul(o.G2_GEN, k)
    img = fmt.write_proving_key(2, 9, [g(3), g(4), None, g(6)], [g(7), g(8), g(9)], [g(20 + i) for i in range(4)], [g(30 + i) for i in range(7)],
                                [h(7), h(8), h(9)], [0, 2, 3, 5], [1, 2, 8], (g(11), g(12), g(13), h(12), h(13)), bitmap_prefix=True)
    info = pkg.gnark_pk_scan(img)
    assert info["bitmap_prefix"] == 1 and info["domain_flag"] == 1 and info["K"] == 7
    # CardinalityInv off by one
    ci = int.from_bytes(img[8:40], "big")
    broken = img[:8] + ((ci + 1) % R).to_bytes(32, "big") + img[40:]
    with pytest.raises(pkg.DvnError) as e:
        pkg.gnark_pk_scan(broken)
    assert "CardinalityInv" in str(e.value)
    # the coset generator of another curve
    off = 8 + 3 * 32
    broken = img[:off] + (22).to_bytes(32, "big") + pow(22, -1, R).to_bytes(32, "big") + img[off + 64:]
    with pytest.raises(pkg.DvnError) as e:
        pkg.gnark_pk_scan(broken)
    assert "coset generator" in str(e.value)


def test_writer_three_bit_encodings(pkg):
    """The oracle's own 3-bit flag encodings on both curves: smallest / largest differ in exactly the sort bit, infinities."""
    from oracle import gnark_key_format_py as fmt
    for curve in ("bls12377", "bw6761"):
        F, S = fmt.suite_format(curve), _suite(curve)
        pt = S.g1_mul(S.g1_gen, 12345)
        neg = (pt[0], S.p - pt[1])
        b, bn = F.g1_bytes(pt, False), F.g1_bytes(neg, False)
        assert len(b) == F.cb and (b[0] ^ bn[0]) == 0x20 and b[0] & 0x80 and b[1:] == bn[1:]
        assert int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big") == pt[0]
        assert F.g1_bytes(None, False)[0] == 0xC0 and F.g1_bytes(None, True)[0] == 0x40 and len(F.g1_bytes(None, True)) == 2 * F.cb
        q = S.g2_mul(S.g2_gen, 77)
        assert len(F.g2_bytes(q, False)) == (2 * F.cb if F.quadratic else F.cb)


@pytest.mark.gpu
@pytest.mark.parametrize("curve,raw,n_c", [("bls12377", False, 700), ("bls12377", True, 300), ("bw6761", False, 500), ("bw6761", True, 200),
                                           ("bls12377", False, 1 << 16)])
def test_key_file_round_trip_proves_on_other_curves(ctx, pkg, tmp_path, curve, raw, n_c):
    """Setup -> export -> WRITE the gnark layout of the curve -> dvn_<curve>_groth16_pk_load_file -> prove from the solver's
    output: the original key's proof and the trapdoor closed form, bit for bit.  Square roots on the device: Tonelli-Shanks on
    BLS12-377's base field (two-adicity 46), the Fp2 complex method over it for G2, a^((p+1)/4) on BW6-761.  The 2^16 case
    takes the expanded-table branch of the loader on every slice."""
    from oracle import gnark_key_format_py as fmt
    from oracle import curves_py as cp
    F, S = fmt.suite_format(curve), _suite(curve)
    n_public = 3
    cs = pkg.R1CS(3, n_c, n_public, curve=curve)
    key = pkg.Groth16Key(ctx, cs, 9)
    arrays = key.export_arrays(cs.n_wires, cs.n_public)
    A, B, Z, K, B2, consts = F.key_arrays_to_points(arrays)
    img = F.write_proving_key(key.log_n, cs.n_wires, A, B, Z, K, B2, arrays["idxA"], arrays["idxB"], consts, raw=raw)
    digest = hashlib.sha256(img).hexdigest()
    path = tmp_path / digest
    path.write_bytes(img)
    lk = pkg.LoadedKey(ctx, path, cs.n_public, sha256_hex=digest, curve=curve)
    assert lk.log_n == key.log_n
    if n_c >= 1 << 16:
        assert min(lk.sizes[k] for k in ("A", "B", "K", "Z", "B2")) >= 1 << 15
    ocs = cp.SyntheticR1CSOn(S, n_c, n_public, 3)
    w = cs.witness()
    ev = lambda rows: np.array([S.fr_to_limbs(sum(c * ocs.witness[i] for i, c in row) % S.r) for row in rows], dtype=np.uint64)
    r, s = S.fr_at(1, 0), S.fr_at(1, 1)
    rl, sl = np.array(S.fr_to_limbs(r), dtype=np.uint64), np.array(S.fr_to_limbs(s), dtype=np.uint64)
    got = lk.prove_solved(w, ev(ocs.A), ev(ocs.B), ev(ocs.C), rl, sl)
    want = key.prove(w, rl, sl)
    for k in ("Ar", "Bs", "Krs"):
        assert (got[k] == want[k]).all(), k
    closed = cp.groth16_expected_from_trapdoor_on(ocs, 9, r, s)
    assert list(got["Ar"]) == S.g1_to_limbs(closed["Ar"]) and list(got["Bs"]) == S.g2_to_limbs(closed["Bs"]) and list(got["Krs"]) == S.g1_to_limbs(closed["Krs"])
    # cache: same content + same arguments -> the resident key; other arguments are refused, not silently served; evict frees it
    again = pkg.LoadedKey(ctx, path, cs.n_public, sha256_hex=digest, curve=curve)
    assert again.h.value == lk.h.value
    with pytest.raises(pkg.DvnError) as e:
        pkg.LoadedKey(ctx, path, cs.n_public, committed_wires=[cs.n_wires - 1], sha256_hex=digest, curve=curve)
    assert "evict" in str(e.value)
    with pytest.raises(pkg.DvnError):                          # a BN254 loader does not take this file
        pkg.LoadedKey(ctx, path, cs.n_public)
    lk.evict()
    fresh = pkg.LoadedKey(ctx, path, cs.n_public, sha256_hex=digest, curve=curve)
    got2 = fresh.prove_solved(w, ev(ocs.A), ev(ocs.B), ev(ocs.C), rl, sl)
    assert (got2["Krs"] == want["Krs"]).all()
    # a damaged compressed point (x moved off the curve) is reported by the device decoder
    if not raw:
        frb = F.frb
        off = 8 + 5 * frb + 1 + 3 * F.cb + 4
        seen = False
        for delta in range(1, 8):
            bad = bytearray(img); bad[off + F.cb - 1] = (img[off + F.cb - 1] + delta) & 0xFF
            try:
                k2 = pkg.LoadedKey(ctx, bytes(bad), cs.n_public, curve=curve); k2.free()
            except pkg.DvnError as e2:
                assert "not on the curve" in str(e2); seen = True; break
        assert seen
    fresh.evict(); key.free(); cs.free()
