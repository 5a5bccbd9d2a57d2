import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
curve = sys.argv[2] if len(sys.argv) > 2 else "bls12377"
n = (1 << log_n) - 5
t = time.time(); cs = pkg.R1CS(3, n, 3, curve=curve); t_cs = time.time() - t
t = time.time(); key = pkg.Groth16Key(ctx, cs, 9); t_set = time.time() - t
dw = ctx.to_device(cs.witness())
rs = np.array([[5, 0, 0, 0], [7, 0, 0, 0]], dtype=np.uint64)
key.prove(dw, rs[0], rs[1])
t = time.time()
for _ in range(5):
    key.prove(dw, rs[0], rs[1])
dt = (time.time() - t) / 5
ms = key.last_ms()
print(f"groth16 {curve} n=2^{log_n}: r1cs {t_cs:.2f}s setup {t_set:.2f}s prove wall {dt*1e3:.1f} ms (device: h ready after {ms['compute_h']:.2f} ms, then {ms['msm']:.2f} ms) -> {1/dt:.2f} proofs/s; sizes {key.sizes}", flush=True)
