"""Synthetic Groth16 at n = 2^20 on the curves given on the command line (bn254 bls12377 bls12381 bw6761): ms per proof."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
log_n = 20
for curve in sys.argv[1:] or ["bls12381", "bw6761"]:
    cs = pkg.R1CS(7, 1 << log_n, 9, curve=curve); key = pkg.Groth16Key(ctx, cs, 3)
    dw = ctx.to_device(cs.witness())
    fw = 6 if curve == "bw6761" else 4
    rs = np.zeros((2, fw), dtype=np.uint64); rs[0, 0], rs[1, 0] = 5, 7
    key.prove(dw, rs[0], rs[1])
    t0 = time.perf_counter()
    for _ in range(5):
        key.prove(dw, rs[0], rs[1])
    dt = (time.perf_counter() - t0) / 5
    print(f"{curve} n=2^{log_n}: {dt*1e3:.2f} ms per proof, sizes {key.sizes}", flush=True)
    ctx.free(dw); key.free(); cs.free()
