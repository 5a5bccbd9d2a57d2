"""Differential sweep of the synthetic Groth16 prover against the oracle's trapdoor closed form at random constraint counts
(not powers of two), public-input counts and blinding factors; every few cases also the shares of worlds 2..5 combined.
Usage: python tests/sweeps/sweep_groth16.py [seconds [seed]]"""
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
mont = lambda x: np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
t_end = time.time() + budget
t_note = time.time() + 30                       # a progress line every half minute: a silent run looks hung to the harness
cases = 0
while time.time() < t_end:
    if time.time() > t_note:
        print(f"{cases} cases so far", flush=True); t_note = time.time() + 30
    nc = int(rng.choice([2, 3, 4, 5, 17, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 3000, 4097, 9000, 20011, 40000]))
    npub = int(rng.integers(1, min(nc, 24)))
    cs_seed, key_seed = int(rng.integers(1, 1 << 20)), int(rng.integers(1, 1 << 20))
    r, s = mont(int(rng.integers(0, 1 << 62)) * int(rng.integers(1, 1 << 62))), mont(int(rng.integers(0, 1 << 62)))
    cs = pkg.R1CS(cs_seed, nc, npub); key = pkg.Groth16Key(ctx, cs, key_seed)
    w = cs.witness(); d = ctx.to_device(w)
    _, want = orc.groth16_synthetic_expected(cs_seed, nc, npub, key_seed, r, s)
    pr = key.prove(d, r, s)
    if not all((pr[k] == want[k]).all() for k in want):
        print(f"MISMATCH prove nc={nc} npub={npub} cs_seed={cs_seed} key_seed={key_seed}", flush=True); sys.exit(1)
    cases += 1
    if cases % 3 == 0:
        world = int(rng.integers(2, 6))
        parts = np.stack([key.prove_shard(d, rk, world) for rk in range(world)])
        pr = key.prove_combine(parts, r, s)
        if not all((pr[k] == want[k]).all() for k in want):
            print(f"MISMATCH shards nc={nc} npub={npub} cs_seed={cs_seed} key_seed={key_seed} world={world}", flush=True); sys.exit(1)
        cases += 1
    ctx.free(d); key.free(); cs.free()
print(f"OK: {cases} cases, no mismatch", flush=True)
ctx.close()
