"""Free device memory before / after repeated create-use-free cycles of every handle type (hipMemGetInfo through torch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
def cycle(k):
    n = 1 << 16
    pts = ctx.synthetic_g1(k, n); d = ctx.synthetic_scalars(1, k & 1, n); ctx.msm_g1(pts, d, n=n); ctx.free(d); pts.free()
    q = ctx.synthetic_g2(k, 1 << 14); d = ctx.synthetic_scalars(1, 0, 1 << 14); ctx.msm_g2(q, d, n=1 << 14); ctx.free(d); q.free()
    dom = ctx.domain(14); x = np.zeros((1 << 14, 4), dtype=np.uint64); ctx.ntt(dom, x, False, 0); dom.free()
    for curve in ("bn254", "bls12377", "bw6761"):
        cs = pkg.R1CS(3, 3000, 2, curve=curve); key = pkg.Groth16Key(ctx, cs, 9)
        fw = 6 if curve == "bw6761" else 4
        rs = np.zeros((2, fw), dtype=np.uint64); rs[0, 0], rs[1, 0] = 5, 7
        key.prove(cs.witness(), rs[0], rs[1]); key.free(); cs.free()
GOLDEN = os.path.join(ROOT, "tests", "golden")
BLOB = bytes.fromhex(open(os.path.join(GOLDEN, "kzg_blobdata1.txt")).read().strip())
_cycle_core = cycle
def cycle(k):
    _cycle_core(k)
    # the handle types added later: KZG setup, committed / imported keys, batched MSM
    ks = pkg.KzgSetup.from_file(ctx, os.path.join(GOLDEN, "kzg_trusted_setup.txt"))
    ks.compute_commitment(BLOB); ks.compute_proof(BLOB, 1234 + k); ks.compute_cell_proofs(BLOB); ks.close()
    cs = pkg.R1CS(3, 3000, 9, n_committed=100); key = pkg.Groth16Key(ctx, cs, 9)
    rs = np.zeros((2, 4), dtype=np.uint64); rs[0, 0], rs[1, 0] = 5, 7
    key.prove_committed(cs.witness(), rs[0], rs[1]); key.free(); cs.free()
    cs = pkg.R1CS(3, 3000, 3); key = pkg.Groth16Key(ctx, cs, 9)
    ik = pkg.ImportedKey(ctx, key.log_n, cs.n_wires, key.export_arrays(cs.n_wires, cs.n_public))
    z = np.zeros((3000, 4), dtype=np.uint64)
    ik.prove_solved(cs.witness(), z, z, z, rs[0], rs[1]); ik.free(); key.free(); cs.free()
    pts = ctx.synthetic_g1(k, 2000); ctx.msm_g1_batched(pts, np.zeros((3, 2000, 4), dtype=np.uint64)); pts.free()
cycle(0); cycle(1)                      # warm up: workspaces of the lanes grow to their final size
torch.cuda.synchronize()
f0 = torch.cuda.mem_get_info()[0]
for k in range(12):
    cycle(k)
torch.cuda.synchronize()
f1 = torch.cuda.mem_get_info()[0]
print(f"free before {f0 >> 20} MiB, after 12 cycles {f1 >> 20} MiB, delta {(f0 - f1) >> 20} MiB")
sys.exit(1 if f0 - f1 > (64 << 20) else 0)
