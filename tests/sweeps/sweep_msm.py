"""Differential sweep of the G1 / G2 MSM against the C oracle at sizes around the tile, segment and policy boundaries, with
scalar vectors built to stress the signed-digit carries (every window at half, r - 1, zeros, one hot bucket).
Usage: python tests/sweeps/sweep_msm.py [seconds [seed]]   (default 60; stops after the time budget, prints the number of cases)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc, bn254_py as o
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = pkg.Context(0)
R = o.R
def mont(v):
    return np.array(o.limbs(o.to_mont(v % R, R)), dtype=np.uint64)
def special(c):
    half = sum(1 << (c * w + c - 1) for w in range(254 // c + 1))
    return [R - 1, half % R, (half - 1) % R, (1 << 253), 1, 0, (1 << c) - 1, 1 << (c - 1), (1 << (c - 1)) + 1]
sizes = sorted({max(1, b + d) for b in (1, 64, 128, 1 << 10, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18) for d in (-1, 0, 1)} | {3, 100003, 50021})
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t_end = time.time() + budget
cases = 0
for rnd in range(1000):
    for n in sizes:
        if time.time() > t_end: break
        seed = int(rng.integers(1, 1 << 30))
        ph = orc.g1_synthetic_points(seed, n)
        if n > 4:                                      # repeated, opposite and infinite points in the set
            ph[1] = ph[0]; ph[2] = ph[0]
            y = sum(int(ph[0][4 + k]) << (64 * k) for k in range(4))          # y -> p - y on the Montgomery image
            ph[2][4:8] = [((o.P - y) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
            ph[3] = 0
        pts = ctx.upload_g1(ph)
        kind = int(rng.integers(0, 5))
        if kind <= 1: sc = orc.scalars(seed, kind, n)
        elif kind == 2:                                # special values scattered over a uniform vector
            sc = orc.scalars(seed, 0, n)
            sp = special(int(rng.choice([13, 15, 16, 17, 20])))
            for i in rng.integers(0, n, size=min(n, 64)): sc[int(i)] = mont(sp[int(rng.integers(0, len(sp)))])
        elif kind == 3:                                # one value everywhere: a single hot bucket per window
            sp = special(int(rng.choice([15, 16, 17, 20])))
            sc = np.repeat(mont(sp[int(rng.integers(0, len(sp)))])[None, :], n, axis=0)
        else:                                          # sparse: mostly zeros
            sc = orc.scalars(seed, 0, n); sc[rng.random(n) < 0.9] = 0
        want = orc.msm_g1(ph, sc)
        got = orc.g1_jac_to_affine(ctx.msm_g1(pts, sc))
        if not (got == want).all():
            print(f"MISMATCH g1 n={n} seed={seed} kind={kind} (default policy)", flush=True); sys.exit(1)
        cases += 1
        if n >= 2048 and rnd % 2 == 0:
            c = int(rng.integers(11, 22))
            pts.set_expanded(c)
            got = orc.g1_jac_to_affine(ctx.msm_g1(pts, sc))
            if not (got == want).all():
                print(f"MISMATCH g1 n={n} seed={seed} kind={kind} window_bits={c}", flush=True); sys.exit(1)
            cases += 1
        pts.free()
        if n <= (1 << 15) + 1 and rnd % 3 == 0:
            p2h = orc.g2_synthetic_points(seed, n); p2 = ctx.upload_g2(p2h)
            if not (orc.g2_jac_to_affine(ctx.msm_g2(p2, sc)) == orc.msm_g2(p2h, sc)).all():
                print(f"MISMATCH g2 n={n} seed={seed} kind={kind}", flush=True); sys.exit(1)
            p2.free(); cases += 1
    if time.time() > t_end: break
    print(f"round {rnd}: {cases} cases so far", flush=True)
print(f"OK: {cases} cases, no mismatch", flush=True)
ctx.close()
