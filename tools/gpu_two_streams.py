"""Does a second stream hide the sort / latency-bound tail of one MSM under the accumulate of another?
Two contexts (= two streams + workspaces) on one GPU, one MSM loop per Python thread (ctypes drops the GIL)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log_n
a, b = pkg.Context(0), pkg.Context(0)
pts = a.synthetic_g1(0xDA71C1, n)
d = a.synthetic_scalars(1, 0, n)
K = 20
for c in (a, b):
    c.msm_g1(pts, d, n=n)
t = time.time()
for _ in range(K): a.msm_g1(pts, d, n=n)
single = (time.time() - t) / K
def loop(c):
    for _ in range(K): c.msm_g1(pts, d, n=n)
ths = [threading.Thread(target=loop, args=(c,)) for c in (a, b)]
t = time.time()
for th in ths: th.start()
for th in ths: th.join()
dual = (time.time() - t) / (2 * K)
print(f"2^{log_n}: single stream {single*1e3:.2f} ms/MSM ({n/single/1e6:.0f} Mscalar/s); two streams {dual*1e3:.2f} ms/MSM ({n/dual/1e6:.0f} Mscalar/s)")
