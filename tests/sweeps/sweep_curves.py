"""Differential sweep of the BLS12-377 / BLS12-381 / BW6-761 instantiations against the big-integer oracle
(oracle/curves_py.py): NTT in all eight modes and ragged computeH at log n = 3..10 with random seeds, and G1 MSMs over
device-generated progressions against the closed form sum s_i (P0 + i D) = (sum s_i) P0 + (sum i s_i) D at 2^10..2^15 points.
Usage: python tests/sweeps/sweep_curves.py [seconds [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import curves_py
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
ctx = pkg.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
suites = [("bls12-377", curves_py.BLS12_377, pkg.Bls12377(ctx)), ("bls12-381", curves_py.BLS12_381, pkg.Bls12381(ctx)),
          ("bw6-761", curves_py.BW6_761, pkg.Bw6761(ctx))]
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    for name, S, dev in suites:
        wide = name == "bw6-761"
        sw = 6 if wide else 4                                               # u64 words per scalar
        # ---- NTT + computeH
        log_n = int(rng.integers(3, 11)); n = 1 << log_n
        seed = int(rng.integers(1, 1 << 30))
        dom = dev.domain(log_n); d = S.domain(n)
        x = [S.fr_at(seed, i) for i in range(n)]
        if rng.random() < 0.3:
            for i in rng.integers(0, n, size=n // 4): x[int(i)] = int(rng.choice([0, 1, 2])) if rng.random() < 0.7 else S.r - 1 - int(rng.integers(0, 3))
        xm = np.array([S.fr_to_limbs(v) for v in x], dtype=np.uint64)
        inv, dec, cos = bool(rng.integers(0, 2)), int(rng.integers(0, 2)), bool(rng.integers(0, 2))
        y = list(x); (d.fft_inverse if inv else d.fft)(y, dec, cos)
        ym = xm.copy(); dev.ntt(dom, ym, inv, dec, cos)
        if [S.fr_from_limbs(r) for r in ym] != y:
            print(f"MISMATCH {name} ntt log_n={log_n} seed={seed} inverse={inv} decimation={dec} coset={cos}", flush=True); sys.exit(1)
        m = int(rng.integers(1, n + 1))
        a, b, c = ([S.fr_at(seed + s, i) for i in range(m)] for s in (1, 2, 3))
        L = lambda v: np.array([S.fr_to_limbs(t) for t in v], dtype=np.uint64)
        got = dev.compute_h(dom, L(a), L(b), L(c))
        if [S.fr_from_limbs(r) for r in got] != d.compute_h(a, b, c):
            print(f"MISMATCH {name} compute_h log_n={log_n} m={m} seed={seed}", flush=True); sys.exit(1)
        dom.free(); cases += 2
        # ---- G1 MSM, closed form
        ln = int(rng.integers(10, 16)); n = 1 << ln; kind = int(rng.integers(0, 2))
        if wide:
            base = S.some_point(1)
            dp = dev.synthetic(1, np.array(S.to_limbs(base), dtype=np.uint64), seed, n)
            ref = S.progression(base, seed, 2); add, neg, mul, tl = S.add, S.neg, S.mul, S.to_limbs
            rinv = pow(1 << 384, -1, S.r)
        else:
            dp = dev.synthetic_g1(seed, n)
            ref = S.synthetic_g1_points(seed, 2); add, neg, mul, tl = S.g1_add, S.g1_neg, S.g1_mul, S.g1_to_limbs
            rinv = pow(1 << 256, -1, S.r)
        p0, dd = ref[0], add(ref[1], neg(ref[0]))
        ds = dev.synthetic_scalars(seed, kind, n)
        sm = np.empty((n, sw), dtype=np.uint64); ctx.d2h(sm, ds)
        so = sm.astype(object)
        mont = sum(so[:, k] << (64 * k) for k in range(sw))
        vals = np.array([int(v) * rinv % S.r for v in mont], dtype=object)                # canonical integer scalars
        s0 = int(vals.sum()); s1 = int((vals * np.arange(n, dtype=object)).sum())
        want = add(mul(p0, s0), mul(dd, s1))
        j = dev.msm(dp, ds, n=n) if wide else dev.msm_g1(dp, ds, n=n)
        w = len(j) // 3
        got = j[:2 * w] if j[2 * w:].any() else np.zeros(2 * w, dtype=np.uint64)
        if list(got) != (tl(want) if want is not None else [0] * (2 * w)):
            print(f"MISMATCH {name} msm n=2^{ln} seed={seed} kind={kind}", flush=True); sys.exit(1)
        ctx.free(ds); dp.free(); cases += 1
    print(f"{cases} cases so far", flush=True)
print(f"OK: {cases} cases, no mismatch", flush=True)
ctx.close()
