"""Blob KZG timings on one MI355X: what BlobEvalData.Set costs per state transition
(/root/reference/crypto/blobs/blob.go:40-90: commitment + 128 cell proofs + proof at z)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
GOLDEN = os.path.join(ROOT, "tests", "golden")
t = time.time(); s = pkg.KzgSetup.from_file(ctx, os.path.join(GOLDEN, "kzg_trusted_setup.txt"))
print(f"setup load (parse + decompress 8192 points on the device): {(time.time()-t)*1e3:.1f} ms", flush=True)
blob = bytes.fromhex(open(os.path.join(GOLDEN, "kzg_blobdata1.txt")).read().strip())
def best(f, reps=10):
    b = 1e9
    for _ in range(reps):
        t = time.time(); f(); b = min(b, time.time() - t)
    return b * 1e3
print(f"compute_commitment : {best(lambda: s.compute_commitment(blob)):.3f} ms (device {ctx.last_kernel_ms(0):.3f})", flush=True)
print(f"compute_proof      : {best(lambda: s.compute_proof(blob, 123456789)):.3f} ms (device MSM {ctx.last_kernel_ms(0):.3f})", flush=True)
print(f"evaluate_barycentric: {best(lambda: s.evaluate_barycentric(blob, 123456789)):.3f} ms", flush=True)
print(f"compute_cell_proofs: {best(lambda: s.compute_cell_proofs(blob), 5):.3f} ms (device MSM {ctx.last_kernel_ms(0):.3f}, accumulate {ctx.last_kernel_ms(1):.3f})", flush=True)
def full():
    s.compute_commitment(blob); s.compute_cell_proofs(blob); s.compute_proof(blob, 123456789)
print(f"BlobEvalData.Set equivalent (commitment + 128 cell proofs + proof): {best(full, 5):.3f} ms", flush=True)
s.close(); ctx.close()
