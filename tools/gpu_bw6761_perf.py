import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0); bw = pkg.Bw6761(ctx)
# smallest x >= 2 with x^3 - 1 a square mod p (p = 3 mod 4), no oracle import: plain integers
p = 0x122E824FB83CE0AD187C94004FAFF3EB926186A81D14688528275EF8087BE41707BA638E584E91903CEBAFF25B423048689C8ED12F9FD9071DCD3DC73EBFF2E98A116C25667A8F8160CF8AEEAF0A437E6913E6870000082F49D00000000008B
x = 2
while True:
    rhs = (x ** 3 - 1) % p; y = pow(rhs, (p + 1) // 4, p)
    if y * y % p == rhs: break
    x += 1
m = lambda v: [((v << 768) % p >> (64 * i)) & (2 ** 64 - 1) for i in range(12)]
base = np.array(m(x) + m(y), dtype=np.uint64)
for log_n in (16, 20, 22, 24):
    n = 1 << log_n
    pts = bw.synthetic(1, base, 0xDA71C1, n); d = bw.synthetic_scalars(1, 0, n)
    for rep in range(3):
        t = time.time(); r = bw.msm(pts, d, n=n); dt = time.time() - t
    print(f"bw6-761 msm 2^{log_n}: wall {dt*1e3:.2f} ms device {ctx.last_kernel_ms(0):.2f} accumulate {ctx.last_kernel_ms(1):.2f} -> {n/dt/1e6:.1f} Mscalar/s", flush=True)
    ctx.free(d); pts.free()
