"""Host-input proving (gnark's solver output in pageable host memory -> proof): dvn_groth16_prove_solved vs the
device-resident-witness prover on the same key (what PCIe costs per proof)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cs = pkg.R1CS(7, 1 << log_n, 9); key = pkg.Groth16Key(ctx, cs, 3)
arr = key.export_arrays(cs.n_wires, cs.n_public)
ik = pkg.ImportedKey(ctx, key.log_n, cs.n_wires, arr)
w = cs.witness()
rng = np.random.default_rng(1)
a = w[9:9 + (1 << log_n)].copy(); b = w[5:5 + (1 << log_n)].copy(); c = w[1:1 + (1 << log_n)].copy()   # any field elements: timing only
r = np.zeros(4, dtype=np.uint64); s = np.zeros(4, dtype=np.uint64); r[0], s[0] = 5, 7
d = ctx.to_device(w)
def best(f, reps=5):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t = min(t, time.perf_counter() - t0)
    return t * 1e3
print(f"n=2^{log_n}: prove (witness resident in HBM, solve on device) {best(lambda: key.prove(d, r, s)):.2f} ms; "
      f"prove_solved (w, a, b, c = {4 * w.nbytes / 2**20:.0f} MiB from pageable host memory) {best(lambda: ik.prove_solved(w, a, b, c, r, s)):.2f} ms", flush=True)
