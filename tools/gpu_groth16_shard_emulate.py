"""Per-rank cost of the row-sharded Groth16 proof at world sizes 1..8, emulated on ONE GPU: this process plays rank 0 of N
(its rows of the five MSMs + the replicated solve/computeH); no all-gather."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
cs = pkg.R1CS(7, 1 << 20, 9); key = pkg.Groth16Key(ctx, cs, 3)
dw = ctx.to_device(cs.witness())
rs = np.zeros((2, 4), dtype=np.uint64); rs[0, 0], rs[1, 0] = 5, 7
for world in (1, 2, 4, 8):
    key.prove_shard(dw, 0, world)
    t = time.time()
    for _ in range(10):
        p = key.prove_shard(dw, 0, world)
    dt = (time.time() - t) / 10
    t = time.time(); key.prove_combine(np.stack([p] * world), rs[0], rs[1]); tc = time.time() - t
    print(f"world {world}: rank-0 shard {dt*1e3:.2f} ms (+ host combine {tc*1e3:.2f} ms) -> {1/(dt+tc):.0f} proofs/s if the all-gather were free", flush=True)
