"""Concurrency sweep: eight host threads call one context at random -- G1 / G2 MSMs (plain, expanded, static), NTTs,
computeH, proofs from ONE shared key and from a committed key -- and every result is compared with the value the same call
returned single-threaded (itself checked against the oracle first).  The header promises that every function is thread-safe.
Only synchronous entry points: outstanding tickets of the asynchronous ones take lanes away from them by contract.
Usage: python tests/sweeps/sweep_threads.py [seconds [seed]]"""
import os, sys, time, threading
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

n1, n2 = 40000, 3000
ph = orc.g1_synthetic_points(61, n1); pts = ctx.upload_g1(ph)            # expanded by the default policy
small = ctx.upload_g1(ph[:900])                                          # plain set
p2h = orc.g2_synthetic_points(5, n2); p2 = ctx.upload_g2(p2h)
scal = [orc.scalars(100 + i, i % 2, n1) for i in range(4)]
dsc = [ctx.to_device(s) for s in scal]
want1 = [orc.msm_g1(ph, s) for s in scal]
want_small = [orc.msm_g1(ph[:900], s[:900]) for s in scal]
want2 = [orc.msm_g2(p2h, s[:n2]) for s in scal]
dom = ctx.domain(12); x = orc.scalars(2, 0, 1 << 12); wx = orc.ntt(x, True, 1, True)
a, b, c = orc.scalars(3, 0, 4000), orc.scalars(4, 0, 4000), orc.scalars(5, 0, 4000)
wh = orc.compute_h(a, b, c, 12)
cs = pkg.R1CS(7, 5000, 9); key = pkg.Groth16Key(ctx, cs, 3); w = cs.witness(); dw = ctx.to_device(w)
r, s = mont(1234567), mont(7654321)
_, wantp = orc.groth16_synthetic_expected(7, 5000, 9, 3, r, s)
csc = pkg.R1CS(9, 3000, 9, n_committed=200); keyc = pkg.Groth16Key(ctx, csc, 5); wc = csc.witness()
wantc = keyc.prove_committed(wc, r, s)
refc = o.SyntheticR1CSCommitted(3000, 9, 200, 9); ec = o.groth16_committed_expected(refc, 5, 1234567, 7654321)
assert list(wantc["Krs"]) == o.g1_to_limbs(ec["Krs"]) and list(wantc["Commitment"]) == o.g1_to_limbs(ec["Commitment"])

errors, counts = [], [0] * 8
t_end = time.time() + budget
def worker(i):
    rng = np.random.default_rng((int(sys.argv[2]) if len(sys.argv) > 2 else 1000) + i)
    try:
        while time.time() < t_end and not errors:
            op = int(rng.integers(0, 8)); k = int(rng.integers(0, 4))
            if op == 0: ok = (orc.g1_jac_to_affine(ctx.msm_g1(pts, scal[k])) == want1[k]).all()
            elif op == 1: ok = (orc.g1_jac_to_affine(ctx.msm_g1(pts, dsc[k], n=n1)) == want1[k]).all()
            elif op == 2: ok = (orc.g1_jac_to_affine(ctx.msm_g1(small, scal[k][:900])) == want_small[k]).all()
            elif op == 3: ok = (orc.g2_jac_to_affine(ctx.msm_g2(p2, scal[k][:n2])) == want2[k]).all()
            elif op == 4: ok = (orc.g1_jac_to_affine(ctx.msm_g1_static(pts, dsc[k], n1)) == want1[k]).all()
            elif op == 5:
                y = x.copy(); ctx.ntt(dom, y, True, 1, True); ok = (y == wx).all()
                ok = ok and (ctx.compute_h(dom, a, b, c) == wh).all()
            elif op == 6:
                pr = key.prove(dw, r, s); ok = all((pr[q] == wantp[q]).all() for q in wantp)
            else:
                pr = keyc.prove_committed(wc, r, s); ok = all((pr[q] == wantc[q]).all() for q in ("Ar", "Bs", "Krs", "Commitment", "CommitmentPok"))
            if not ok: errors.append((i, op, k))
            counts[i] += 1
    except Exception as e:                                       # noqa: BLE001
        errors.append((i, repr(e)))
ths = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
for th in ths: th.start()
while any(th.is_alive() for th in ths):
    time.sleep(10); print(f"{sum(counts)} calls so far", flush=True)
for th in ths: th.join()
if errors:
    print("FAILED:", errors[:5], flush=True); sys.exit(1)
print(f"OK: {sum(counts)} concurrent calls from 8 threads, all results equal to the single-threaded ones", flush=True)
