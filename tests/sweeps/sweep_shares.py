"""Differential sweep of the shared MSM (`dvn_msm_g1_share_*`): random sizes, worlds 1..16 and window sizes; the sum of the
shares must equal the C oracle's MSM.  Covers plain sets (whole-window shares) and expanded sets (entry-count shares).
Usage: python tests/sweeps/sweep_shares.py [seconds [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = pkg.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
t_end = time.time() + budget
t_note = time.time() + 30                       # a progress line every half minute: a silent run looks hung to the harness
cases = 0
while time.time() < t_end:
    if time.time() > t_note:
        print(f"{cases} cases so far", flush=True); t_note = time.time() + 30
    n = int(rng.choice([1, 2, 7, 100, 1000, 1023, 1024, 1025, 5000, 8191, 8193, 20000, 40001, 70000]))
    seed = int(rng.integers(1, 1 << 30)); kind = int(rng.integers(0, 2))
    ph = orc.g1_synthetic_points(seed, n); pts = ctx.upload_g1(ph)
    sc = orc.scalars(seed, kind, n); d = ctx.to_device(sc)
    want = orc.msm_g1(ph, sc)
    for c in (None, int(rng.integers(11, 22)) if n >= 1024 else None):
        if c is not None: pts.set_expanded(c)
        world = int(rng.integers(1, 17))
        parts = []
        for lo in range(0, world, 4):                  # at most four shares in flight (the context's lanes)
            tk = [ctx.msm_g1_share_submit(pts, d, n, r, world) for r in range(lo, min(world, lo + 4))]
            parts += [ctx.msm_g1_share_collect(t) for t in tk]
        got = orc.g1_jac_to_affine(ctx.msm_g1_sum(np.stack(parts)))
        if not (got == want).all():
            print(f"MISMATCH n={n} seed={seed} kind={kind} window_bits={c} world={world}", flush=True); sys.exit(1)
        cases += 1
    ctx.free(d); pts.free()
print(f"OK: {cases} cases, no mismatch", flush=True)
ctx.close()
