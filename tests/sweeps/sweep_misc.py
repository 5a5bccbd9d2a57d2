"""Differential sweep of the entry points around the MSM: `dvn_msm_g1_static` against `dvn_msm_g1` (random sizes, offsets and
window sizes, alternating large and small calls so the scratch always holds another call's data), `dvn_msm_g1_batched`
against single MSMs, and the committed Groth16 prover against the oracle's closed form at random shapes.
Usage: python tests/sweeps/sweep_misc.py [seconds [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc, bn254_py as o
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
ctx = pkg.Context(0)
R = o.R
mont = lambda x: np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
t_end = time.time() + budget
t_note = time.time() + 30                       # a progress line every half minute: a silent run looks hung to the harness
cases = refused = 0
while time.time() < t_end:
    if time.time() > t_note:
        print(f"{cases} cases so far", flush=True); t_note = time.time() + 30
    # ---- the allocation-free MSM
    n = int(rng.choice([1, 2, 100, 1023, 1024, 1025, 4000, 8192, 8193, 30000, 32768, 50000, 65535, 65536]))
    seed = int(rng.integers(1, 1 << 30)); kind = int(rng.integers(0, 2))
    pts = ctx.synthetic_g1(seed, n); d = ctx.synthetic_scalars(seed, kind, n)
    if n >= 2048 and rng.random() < 0.5: pts.set_expanded(int(rng.integers(11, 22)))
    off = int(rng.integers(0, n)) if rng.random() < 0.3 else 0
    m = int(rng.integers(1, n - off + 1)) if off or rng.random() < 0.2 else n
    try:
        got = ctx.msm_g1_static(pts, d, m, off=off)
    except pkg.DvnError as e:                          # refusals are allowed (limits of the fixed scratch), wrong answers are not
        got = None; refused += 1
    if got is not None:
        sc = np.zeros((n, 4), dtype=np.uint64); ctx.d2h(sc, d)
        want = orc.msm_g1(pts.download()[off:off + m], sc[:m])
        if not (orc.g1_jac_to_affine(got) == want).all():
            print(f"MISMATCH static n={n} m={m} off={off} seed={seed} kind={kind}", flush=True); sys.exit(1)
        cases += 1
    ctx.free(d)
    # ---- batched MSM over the same set
    if n <= 8193:
        B = int(rng.integers(1, 6))
        scs = np.stack([orc.scalars(seed + b, int(rng.integers(0, 2)), n) for b in range(B)])
        got = ctx.msm_g1_batched(pts, scs)
        ph = pts.download()
        for b in range(B):
            if not (orc.g1_jac_to_affine(got[b]) == orc.msm_g1(ph, scs[b])).all():
                print(f"MISMATCH batched n={n} B={B} b={b} seed={seed}", flush=True); sys.exit(1)
        cases += 1
    pts.free()
    # ---- committed Groth16
    nc = int(rng.choice([50, 257, 1000, 1024, 4099, 10000]))
    npub = int(rng.integers(2, 12)); ncom = int(rng.integers(1, max(2, nc // 8)))
    cs_seed, key_seed = int(rng.integers(1, 1 << 20)), int(rng.integers(1, 1 << 20))
    r, s = int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 62))
    cs = pkg.R1CS(cs_seed, nc, npub, n_committed=ncom); key = pkg.Groth16Key(ctx, cs, key_seed)
    ref = o.SyntheticR1CSCommitted(nc, npub, ncom, cs_seed)
    want = o.groth16_committed_expected(ref, key_seed, r, s)
    got = key.prove_committed(cs.witness(), mont(r), mont(s))
    ok = (list(got["Commitment"]) == o.g1_to_limbs(want["Commitment"]) and list(got["CommitmentPok"]) == o.g1_to_limbs(want["CommitmentPok"])
          and list(got["challenge"]) == o.limbs(o.to_mont(want["challenge"], R)) and list(got["Ar"]) == o.g1_to_limbs(want["Ar"])
          and list(got["Bs"]) == o.g2_to_limbs(want["Bs"]) and list(got["Krs"]) == o.g1_to_limbs(want["Krs"]))
    if not ok:
        print(f"MISMATCH committed nc={nc} npub={npub} ncom={ncom} cs_seed={cs_seed} key_seed={key_seed}", flush=True); sys.exit(1)
    key.free(); cs.free(); cases += 1
print(f"OK: {cases} cases, no mismatch ({refused} static calls refused)", flush=True)
ctx.close()
