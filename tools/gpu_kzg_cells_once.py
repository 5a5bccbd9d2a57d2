"""One warm-up + 3 cell-proof calls (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
GOLDEN = os.path.join(ROOT, "tests", "golden")
s = pkg.KzgSetup.from_file(ctx, os.path.join(GOLDEN, "kzg_trusted_setup.txt"))
blob = bytes.fromhex(open(os.path.join(GOLDEN, "kzg_blobdata1.txt")).read().strip())
for _ in range(4):
    s.compute_cell_proofs(blob)
for _ in range(4):
    s.compute_proof(blob, 123456789)
s.close(); ctx.close()
