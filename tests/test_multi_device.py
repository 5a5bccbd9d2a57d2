"""Multi-GPU inside the library (the node proves one statetransition at a time,
/root/reference/sequencer/statetransition.go:80): dvn_init honours every device ordinal, one proof is spread over the devices
of a context by entry count, and the partial results are added on the host.  A one-GPU machine rehearses the path with the
same ordinal listed twice or three times (independent contexts sharing the GPU); results must equal the single-device proof
bit for bit.  Also here: the MSM shared by entry count (what bench.py --gpus N runs per rank), the per-proof statistics, the
Pedersen commitment on the voteverifier / aggregator curves, and the error paths around them.
"""
import ctypes as C

import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def mont(o, x):
    return np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)


def test_error_strings_are_thread_independent(pkg):
    L = pkg.load_library()
    assert L.dvn_error_string(0) == b"ok" and L.dvn_error_string(-1) == b"invalid argument" and L.dvn_error_string(-3).startswith(b"no usable")
    assert L.dvn_error_string(-99) == b"unknown error code"
    assert L.dvn_ctx_device_count(None) == 0 and L.dvn_ctx_device_ordinal(None) == -1


def test_msm_sum_is_host_only(pkg, opy, orc):
    """dvn_msm_g1_sum adds G1Jac images without a device: P + Q + infinity + (-Q) = P."""
    o = opy
    L = pkg.load_library()
    jac = lambda pt: np.array(o.g1_to_limbs(pt) + o.limbs(o.to_mont(1, o.P)), dtype=np.uint64)
    p, q = o.g1_mul(o.G1_GEN, 11), o.g1_mul(o.G1_GEN, 22)
    inf = np.concatenate([jac(p)[:8], np.zeros(4, dtype=np.uint64)])
    parts = np.stack([jac(p), jac(q), inf, jac(o.g1_neg(q))])
    out = np.zeros(12, dtype=np.uint64)
    assert L.dvn_msm_g1_sum(parts.ctypes.data, 4, out.ctypes.data) == 0
    assert list(out[:8]) == o.g1_to_limbs(p)
    assert L.dvn_msm_g1_sum(parts.ctypes.data, 0, out.ctypes.data) == 0 and not out[8:].any()     # empty sum: infinity (Z = 0)


@pytest.mark.gpu
def test_dvn_init_honours_every_ordinal(pkg, orc):
    c = pkg.Context([0, 0, 0])
    assert c.device_count() == 3
    subs = [c.device(i) for i in range(3)]
    assert subs[0] is c and all(c.L.dvn_ctx_device_ordinal(s.h) == 0 for s in subs)
    assert len({s.h.value for s in subs}) == 3, "three distinct contexts"
    with pytest.raises(pkg.DvnError):
        c.device(3)
    # every sub-context is a working context of its own (own streams, lanes, workspace, static scratch)
    ph = orc.g1_synthetic_points(71, 700); sc = orc.scalars(3, 1, 700)
    want = orc.msm_g1(ph, sc)
    for s in subs:
        pts = s.upload_g1(ph)
        assert (orc.g1_jac_to_affine(s.msm_g1(pts, sc)) == want).all()
        pts.free()
    c.close()
    with pytest.raises(pkg.DvnError) as e:
        pkg.Context([0, 99])
    assert e.value.code == -3


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shared_msm_by_entry_count(ctx, pkg, orc, world):
    """The shares of all ranks add up to the MSM for any world size; on an expanded set a share is an equal entry range (13
    rows over 8 ranks are not whole rows).  Plain (small) sets: rank 0 computes everything."""
    for n, kind in ((70001, 0), (40000, 1), (900, 0)):
        pts = ctx.synthetic_g1(0xDA71C1, n)
        d = ctx.synthetic_scalars(31, kind, n)
        want = ctx.msm_g1(pts, d, n=n)
        tickets = []
        parts = []
        for rank in range(world):                          # at most four lanes: collect as we go
            tickets.append(ctx.msm_g1_share_submit(pts, d, n, rank, world))
            if len(tickets) == 4:
                parts.append(ctx.msm_g1_share_collect(tickets.pop(0)))
        while tickets:
            parts.append(ctx.msm_g1_share_collect(tickets.pop(0)))
        got = ctx.msm_g1_sum(np.stack(parts))
        assert (got == want).all(), (n, kind, world)
        if n == 900:
            assert all((p[8:] == 0).all() for p in parts[1:]), "small set: only rank 0 contributes"
        ctx.free(d); pts.free()
    pts = ctx.synthetic_g1(1, 10)
    d = ctx.synthetic_scalars(31, 0, 10)
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1_share_submit(pts, d, 10, 2, 2)             # rank outside the world
    ctx.free(d); pts.free()


@pytest.mark.gpu
def test_one_msm_over_the_devices_of_a_context(pkg):
    """dvn_msm_g1_multi on a three-"device" context == the single-device MSM (itself checked against the oracle in
    test_gpu_parity.py); an expanded set (every device runs a real entry range) and a plain one (device 0 computes it whole)."""
    mc = pkg.Context([0, 0, 0])
    devs = [mc.device(i) for i in range(3)]
    for n, kind in ((70001, 0), (40000, 1), (900, 0), ((1 << 19) + 77, 0)):      # the last one: 16 MiB of host scalars, staged upload on every device at once
        sets = [d.synthetic_g1(0xDA71C1, n) for d in devs]
        ds = devs[0].synthetic_scalars(77, kind, n)
        sc = np.empty((n, 4), dtype=np.uint64); devs[0].d2h(sc, ds)
        want = devs[0].msm_g1(sets[0], sc)
        got = pkg.msm_g1_multi(mc, sets, sc)
        assert (got == want).all(), (n, kind)
        if n == 70001:
            got_off = pkg.msm_g1_multi(mc, sets, sc[:50000], off=123)
            assert (got_off == devs[0].msm_g1(sets[0], sc[:50000], off=123)).all()
            with pytest.raises(pkg.DvnError) as e:
                pkg.msm_g1_multi(mc, sets[:2], sc)                   # one set per device
            assert e.value.code == -1
            short = devs[2].synthetic_g1(0xDA71C1, n - 1)
            with pytest.raises(pkg.DvnError) as e:
                pkg.msm_g1_multi(mc, [sets[0], sets[1], short], sc[:10])
            assert "not the same point set" in str(e.value)
            short.free()
        devs[0].free(ds)
        for p in sets:
            p.free()
    mc.close()


@pytest.mark.gpu
def test_one_proof_over_the_devices_of_a_context(pkg, opy, orc):
    """dvn_groth16_prove_multi / _prove_solved_multi on a three-"device" context == the single-device proof == the closed form.
    n = 2^16 - 3: every key set is an expanded table, so every device runs a real entry range."""
    o = opy
    n_c = (1 << 16) - 3
    mc = pkg.Context([0, 0, 0])
    cs = pkg.R1CS(7, n_c, 9)
    keys = [pkg.Groth16Key(mc.device(i), cs, 3) for i in range(3)]
    w = cs.witness()
    r, s = mont(o, 5), mont(o, 6)
    single = keys[0].prove(w, r, s)
    _, closed = orc.groth16_synthetic_expected(7, n_c, 9, 3, r, s)
    multi = pkg.prove_multi(mc, keys, w, r, s)
    for k in ("Ar", "Bs", "Krs"):
        assert (multi[k] == single[k]).all() and (multi[k] == closed[k]).all(), k
    st = keys[2].last_stats()
    assert st["rank"] == 2 and st["world"] == 3 and st["device_ms"]["total"] > 0
    # the gnark-facing form: imported keys, proof from the solver's (w, a, b, c)
    arrays = keys[0].export_arrays(cs.n_wires, cs.n_public)
    iks = [pkg.ImportedKey(mc.device(i), keys[0].log_n, cs.n_wires, arrays) for i in range(3)]
    ref = o.SyntheticR1CS(n_c, 9, 7)
    a, b, c = (np.array([o.limbs(o.to_mont(v, R)) for v in col], dtype=np.uint64) for col in ref.abc())
    solved = pkg.prove_multi(mc, iks, w, r, s, solved=(a, b, c))
    for k in ("Ar", "Bs", "Krs"):
        assert (solved[k] == closed[k]).all(), k
    # error paths: wrong key count, a key from another device's context list, device-resident witness flag is not offered
    with pytest.raises(pkg.DvnError):
        pkg.prove_multi(mc, keys[:2], w, r, s)
    with pytest.raises(pkg.DvnError):
        pkg.prove_multi(mc, iks, w, r, s)                     # imported keys have no constraint system
    # after the failures the contexts still prove (no lane is left busy)
    again = pkg.prove_multi(mc, keys, w, r, s)
    assert (again["Krs"] == closed["Krs"]).all()
    for k in iks + keys:
        k.free()
    cs.free(); mc.close()


@pytest.mark.gpu
def test_small_key_sets_do_not_break_a_shared_proof(ctx, pkg, opy, orc):
    """A key with sets below 2^15 points cannot be cut by entry range: rank 0 computes those MSMs whole, the other ranks
    contribute infinity, and the shares still add up to the proof."""
    cs = pkg.R1CS(7, 1000, 3)
    key = pkg.Groth16Key(ctx, cs, 3)
    r, s = mont(opy, 9), mont(opy, 4)
    ref = key.prove(cs.witness(), r, s)
    parts = np.stack([key.prove_shard(cs.witness(), rank, 3) for rank in range(3)])
    got = key.prove_combine(parts, r, s)
    for k in ("Ar", "Bs", "Krs"):
        assert (got[k] == ref[k]).all(), k
    key.free(); cs.free()


@pytest.mark.gpu
def test_plain_prover_refuses_a_committed_key(ctx, pkg, opy):
    """A key with a BSB22 commitment through the generic prove seam would return a proof that cannot verify (commitment wire
    unset): every plain entry point refuses it, and the context keeps working afterwards."""
    o = opy
    cs = pkg.R1CS(7, 200, 9, n_committed=8)
    key = pkg.Groth16Key(ctx, cs, 3)
    w = cs.witness()
    r, s = mont(o, 1), mont(o, 2)
    out = np.zeros(32, dtype=np.uint64)
    L = ctx.L
    assert L.dvn_groth16_prove(ctx.h, key.h, w.ctypes.data, 0, r.ctypes.data, s.ctypes.data, out.ctypes.data) == -1
    assert b"BSB22 commitment" in L.dvn_last_error(ctx.h)
    part = np.zeros(72, dtype=np.uint64)
    assert L.dvn_groth16_prove_shard(ctx.h, key.h, w.ctypes.data, 0, 0, 2, part.ctypes.data) == -1
    assert L.dvn_groth16_prove_solved(ctx.h, key.h, w.ctypes.data, w.ctypes.data, w.ctypes.data, w.ctypes.data, 10, r.ctypes.data, s.ctypes.data, out.ctypes.data) == -1
    # the prover-package mirror routes such a key to the committed prover (callGPUProver): its proof image continues with
    # Commitments[0] | CommitmentPok, as gnark's Proof does -- dvn_groth16_proof_words says how long it is
    assert key.proof_words() == 48
    L.dvn_prover_set_use_gpu(1)
    L.dvn_prover_set_cpu_backend(None, None)
    L.dvn_prover_set_prover(None, None)
    out48 = np.zeros(48, dtype=np.uint64)
    assert L.dvn_prover_prove(ctx.h, 1, cs.h, key.h, w.ctypes.data, r.ctypes.data, s.ctypes.data, out48.ctypes.data) == 0
    assert L.dvn_prover_last_backend() == 1
    L.dvn_prover_set_use_gpu(0)
    got = key.prove_committed(w, r, s)                              # the entry point itself, before and after
    want = o.groth16_committed_expected(o.SyntheticR1CSCommitted(200, 9, 8, 7), 3, 1, 2)
    assert list(got["Krs"]) == o.g1_to_limbs(want["Krs"])
    assert (out48[0:8] == got["Ar"]).all() and (out48[8:24] == got["Bs"]).all() and (out48[24:32] == got["Krs"]).all()
    assert (out48[32:40] == got["Commitment"]).all() and (out48[40:48] == got["CommitmentPok"]).all()
    key.free(); cs.free()


@pytest.mark.gpu
def test_committed_proof_shared_by_ranks_and_by_devices(pkg, opy):
    """The statetransition shape (8 public inputs + one BSB22 commitment) shared between GPUs: per-rank shares
    (dvn_groth16_prove_committed_shard + _combine, worlds 1, 2, 3, 8 played by one process) and the one-call form over a
    three-"device" context (dvn_groth16_prove_committed_multi) both reproduce the single-device committed proof and the
    oracle's closed form (commitment, PoK, challenge, Ar, Bs, Krs) bit for bit.  2^16 constraints: every large key set is an
    expanded table, so every rank runs a real entry range; the 2^12 commitment bases are replicated."""
    o = opy
    n_c, n_cm = 1 << 16, 1 << 12
    mc = pkg.Context([0, 0, 0])
    cs = pkg.R1CS(7, n_c, 9, n_committed=n_cm)
    keys = [pkg.Groth16Key(mc.device(i), cs, 3) for i in range(3)]
    w = cs.witness()
    r, s = mont(o, 1234567), mont(o, 7654321)
    want = o.groth16_committed_expected(o.SyntheticR1CSCommitted(n_c, 9, n_cm, 7), 3, 1234567, 7654321)
    def check(got, tag):
        assert list(got["Commitment"]) == o.g1_to_limbs(want["Commitment"]), tag
        assert list(got["CommitmentPok"]) == o.g1_to_limbs(want["CommitmentPok"]), tag
        assert list(got["challenge"]) == o.limbs(o.to_mont(want["challenge"], R)), tag
        assert list(got["Ar"]) == o.g1_to_limbs(want["Ar"]) and list(got["Bs"]) == o.g2_to_limbs(want["Bs"]), tag
        assert list(got["Krs"]) == o.g1_to_limbs(want["Krs"]), tag
    check(keys[0].prove_committed(w, r, s), "single")
    from davinci_node_amd import sharding
    for world in (1, 2, 3, 8):
        shares = [keys[0].prove_committed_shard(w, rank, world) for rank in range(world)]
        got = keys[0].prove_combine(np.stack([sh["partial"] for sh in shares]), r, s)
        got.update({k: shares[-1][k] for k in ("Commitment", "CommitmentPok", "challenge")})
        check(got, f"world {world}")
        assert all((sh["Commitment"] == shares[0]["Commitment"]).all() for sh in shares)
    check(sharding.sharded_groth16_prove_committed(keys[0], w, r, s, 0, 1), "helper, world 1")
    check(pkg.prove_committed_multi(mc, keys, w, r, s), "multi")
    # a plain key is refused by the committed entry points; the plain multi-device prover names the right call for a committed key
    cs0 = pkg.R1CS(7, 300, 9); k0 = [pkg.Groth16Key(mc.device(i), cs0, 3) for i in range(3)]
    with pytest.raises(pkg.DvnError):
        pkg.prove_committed_multi(mc, k0, cs0.witness(), r, s)
    with pytest.raises(pkg.DvnError) as e:
        pkg.prove_multi(mc, keys, w, r, s)
    assert "dvn_groth16_prove_committed_multi" in str(e.value)
    check(pkg.prove_committed_multi(mc, keys, w, r, s), "multi, after the refusals")
    for k in k0 + keys:
        k.free()
    cs0.free(); cs.free(); mc.close()


@pytest.mark.gpu
def test_stats_json_of_a_proof(ctx, pkg, opy):
    cs = pkg.R1CS(7, 1 << 12, 9)
    key = pkg.Groth16Key(ctx, cs, 3)
    key.prove(cs.witness(), mont(opy, 1), mont(opy, 2))
    st = key.last_stats()
    assert st["suite"] == 0 and st["log_n"] == 12 and st["wires"] == cs.n_wires and st["world"] == 1
    assert set(st["msm"]) == {"G1.A", "G1.B", "G1.K", "G1.Z", "G2.B"}
    assert st["msm"]["G1.Z"]["pairs"] == (1 << 12) - 1 and st["msm"]["G1.K"]["pairs"] == cs.n_wires - cs.n_public
    assert st["bytes"]["host_to_device"] == cs.n_wires * 32
    assert st["device_ms"]["total"] >= st["device_ms"]["until_h_ready"] > 0 and st["achieved_GBps"] > 0
    pairs = sum(v["pairs"] for v in st["msm"].values())
    assert st["bytes"]["algorithmic"] == 96 * (pairs - st["msm"]["G2.B"]["pairs"]) + 160 * st["msm"]["G2.B"]["pairs"] + (7 * 64 + 4 * 32) * (1 << 12)
    need = C.c_size_t()
    assert ctx.L.dvn_groth16_last_stats_json(key.h, C.create_string_buffer(8), 8, C.byref(need)) == -1 and need.value > 8
    key.free(); cs.free()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12377", "bls12381", "bw6761"])
def test_pedersen_commit_on_the_other_curves(ctx, pkg, curve):
    """Row a14 on the voteverifier / aggregator curves (prover_gpu.go:36-57): commitment = MSM(Basis, v), pok = MSM(sigma Basis, v)
    = sigma * commitment, against the big-int oracle."""
    from oracle import curves_py as cp
    n, sigma = 300, 0x1234567890ABCDEF1234567
    if curve == "bw6761":
        S, api = cp.BW6_761, pkg.Bw6761(ctx)
        basis = [S.mul(S.g1_gen, 3 + 5 * i) for i in range(n)]
        up = lambda pts: api.upload(1, np.array([S.to_limbs(p) for p in pts], dtype=np.uint64))
    else:
        S = cp.BLS12_377 if curve == "bls12377" else cp.BLS12_381
        api = pkg.Bls12377(ctx) if curve == "bls12377" else pkg.Bls12381(ctx)
        basis = S.synthetic_g1_points(81, n)
        up = lambda pts: api.upload_g1(np.array([S.g1_to_limbs(p) for p in pts], dtype=np.uint64))
    bes = [S.g1_mul(p, sigma) for p in basis]
    vals = [S.fr_at(9, i) if i % 3 else i % 2 for i in range(n)]           # zeros and ones among uniform values
    v = np.array([S.fr_to_limbs(x) for x in vals], dtype=np.uint64)
    pb, ps = up(basis), up(bes)
    c, k = api.pedersen_commit(pb, ps, v)
    want = S.g1_msm(basis, vals) if curve != "bw6761" else S.msm(basis, vals)
    assert list(c) == S.g1_to_limbs(want)
    assert list(k) == S.g1_to_limbs(S.g1_mul(want, sigma))
    # a BN254 basis is refused
    bn = ctx.synthetic_g1(1, 16)
    with pytest.raises(pkg.DvnError):
        api.pedersen_commit(bn, bn, v[:16])
    bn.free(); pb.free(); ps.free()


@pytest.mark.gpu
def test_handles_of_another_curve_are_refused_by_the_async_and_windowed_entry_points(ctx, pkg):
    """A BLS12-377 G1 handle has wider points: the BN254 submit / window / download calls must refuse it, not read it."""
    api = pkg.Bls12377(ctx)
    other = api.synthetic_g1(3, 64)
    d = ctx.synthetic_scalars(1, 0, 64)
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1_submit(other, d, n=64)
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1_windows(other, d, 0, 1, n=64)
    with pytest.raises(pkg.DvnError):
        other.download()
    with pytest.raises(pkg.DvnError):
        ctx.msm_g1_share_submit(other, d, 64, 0, 1)
    # a ticket of one curve cannot be collected through another curve's collect
    pts = ctx.synthetic_g1(3, 64)
    t = ctx.msm_g1_submit(pts, d, n=64)
    out = np.zeros(18, dtype=np.uint64)
    assert ctx.L.dvn_msm_g2_collect(ctx.h, t, out.ctypes.data) == -1
    ctx.msm_g1_collect(t)
    ctx.free(d); pts.free(); other.free()


@pytest.mark.gpu
@pytest.mark.parametrize("expanded", [False, True])
def test_msm_with_repeated_and_opposite_points(ctx, opy, orc, expanded):
    """Real keys repeat points (A_i = A_j) and may hold P next to -P: with equal scalars they meet in one bucket, so the lazily
    reduced accumulator runs its doubling branch (P + P) and its cancellation branch (P + (-P) = infinity) on the device."""
    base = orc.g1_synthetic_points(5, 64)
    neg = base.copy()
    P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    for i in range(64):                                   # y -> p - y in Montgomery form: negation commutes with the image
        y = sum(int(neg[i, 4 + k]) << (64 * k) for k in range(4))
        y = (P - y) % P
        neg[i, 4:8] = [(y >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    rows = []
    for i in range(64):                                   # P, P, -P, Q, Q, ... with identical scalars inside each group
        rows += [base[i], base[i], neg[i], base[(i + 1) % 64], base[(i + 1) % 64]]
    reps = 120 if expanded else 9                         # >= 2^15 points: the expanded-table path
    ph = np.ascontiguousarray(np.tile(np.array(rows, dtype=np.uint64), (reps, 1)))
    n = ph.shape[0]
    sc_group = orc.scalars(77, 0, n // 5 + 1)
    sc = np.ascontiguousarray(np.repeat(sc_group, 5, axis=0)[:n])
    pts = ctx.upload_g1(ph)
    assert (orc.g1_jac_to_affine(ctx.msm_g1(pts, sc)) == orc.msm_g1(ph, sc)).all()
    ones = np.ascontiguousarray(np.tile(mont(opy, 1), (n, 1)))
    assert (orc.g1_jac_to_affine(ctx.msm_g1(pts, ones)) == orc.msm_g1(ph, ones)).all()      # every point in ONE bucket
    pts.free()


@pytest.mark.gpu
def test_msm_g2_with_repeated_and_opposite_points(ctx, orc):
    base = orc.g2_synthetic_points(5, 32)
    P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    neg = base.copy()
    for i in range(32):
        for c0 in (8, 12):                                # Y.A0, Y.A1
            y = sum(int(neg[i, c0 + k]) << (64 * k) for k in range(4))
            y = (P - y) % P
            neg[i, c0:c0 + 4] = [(y >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    rows = []
    for i in range(32):
        rows += [base[i], base[i], neg[i], base[(i + 1) % 32], base[(i + 1) % 32]]
    for reps in (7, 210):                                 # plain set, expanded set
        ph = np.ascontiguousarray(np.tile(np.array(rows, dtype=np.uint64), (reps, 1)))
        n = ph.shape[0]
        sc = np.ascontiguousarray(np.repeat(orc.scalars(78, 0, n // 5 + 1), 5, axis=0)[:n])
        pts = ctx.upload_g2(ph)
        assert (orc.g2_jac_to_affine(ctx.msm_g2(pts, sc)) == orc.msm_g2(ph, sc)).all(), reps
        pts.free()
