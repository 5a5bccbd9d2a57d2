"""Differential sweep of the NTT (all eight modes) and computeH (ragged lengths) against the C oracle at log n = 1..17.
Usage: python tests/sweeps/sweep_ntt.py [seconds [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = pkg.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
doms = {}
t_end = time.time() + budget
t_note = time.time() + 30                       # a progress line every half minute: a silent run looks hung to the harness
cases = 0
while time.time() < t_end:
    if time.time() > t_note:
        print(f"{cases} cases so far", flush=True); t_note = time.time() + 30
    log_n = int(rng.integers(1, 18)); n = 1 << log_n
    dom = doms.get(log_n) or doms.setdefault(log_n, ctx.domain(log_n))
    seed = int(rng.integers(1, 1 << 30))
    x = orc.scalars(seed, int(rng.integers(0, 2)), n)
    if rng.random() < 0.2: x[rng.random(n) < 0.5] = 0
    inv, dec, cos = bool(rng.integers(0, 2)), int(rng.integers(0, 2)), bool(rng.integers(0, 2))
    y = x.copy(); ctx.ntt(dom, y, inv, dec, cos)                       # in place
    if not (y == orc.ntt(x, inv, dec, cos)).all():
        print(f"MISMATCH ntt log_n={log_n} seed={seed} inverse={inv} decimation={dec} coset={cos}", flush=True); sys.exit(1)
    m = int(rng.integers(1, n + 1))
    a, b, c = orc.scalars(seed + 1, 0, m), orc.scalars(seed + 2, 0, m), orc.scalars(seed + 3, int(rng.integers(0, 2)), m)
    if not (ctx.compute_h(dom, a, b, c) == orc.compute_h(a, b, c, log_n)).all():
        print(f"MISMATCH compute_h log_n={log_n} m={m} seed={seed}", flush=True); sys.exit(1)
    cases += 2
print(f"OK: {cases} cases, no mismatch", flush=True)
ctx.close()
