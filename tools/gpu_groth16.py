"""Statetransition-shaped synthetic Groth16: Setup + Prove timings (single GPU)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc, bn254_py as o
ctx = pkg.Context(0)
R = o.R
mont = lambda x: np.array(o.limbs(o.to_mont(x % R, R)), dtype=np.uint64)
for log_n in [int(a) for a in sys.argv[1:]] or [16, 18, 20]:
    nc = 1 << log_n
    t = time.time(); cs = pkg.R1CS(7, nc, 9); t_cs = time.time() - t
    t = time.time(); key = pkg.Groth16Key(ctx, cs, 3); t_setup = time.time() - t
    w = cs.witness(); d = ctx.to_device(w)
    r, s = mont(1234567), mont(7654321)
    for rep in range(3):
        t = time.time(); pr = key.prove(d, r, s); dt = time.time() - t
    ms = key.last_ms()
    t = time.time(); _, want = orc.groth16_synthetic_expected(7, nc, 9, 3, r, s); t_chk = time.time() - t
    ok = all((pr[k] == want[k]).all() for k in want)
    print(f"groth16 n=2^{log_n}: r1cs {t_cs:.2f}s setup {t_setup:.2f}s prove wall {dt*1e3:.1f} ms (device: h ready after {ms['compute_h']:.2f} ms, then {ms['msm']:.2f} ms) "
          f"-> {1/dt:.2f} proofs/s; sizes {key.sizes}; trapdoor check {'PASS' if ok else 'FAIL'} ({t_chk:.1f}s)", flush=True)
    ctx.free(d); key.free(); cs.free()
ctx.close()
