"""Per-rank cost of the window-sharded MSM at world sizes 1..8, emulated on ONE GPU: this process plays rank 0 of N
(its share of the entries); the all-gather is replaced by a local copy, so the figure is the GPU + host side of a
step without the RCCL latency.  Arguments: how many steps to keep in flight (default 2; several values = several tables).
DVN_EMULATE_WINDOWS="8:17,8:18,4:18": additional lines with the table re-expanded for that window size at that world."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
import davinci_node_amd.sharding as sh


class FakeDist:
    def all_gather_into_tensor(self, recv, send):
        w = recv.shape[0] // send.shape[0]
        for r in range(w):
            recv[r * send.shape[0]:(r + 1) * send.shape[0]] = send


ctx = pkg.Context(0)
n = 1 << 22
pts = ctx.synthetic_g1(0xDA71C1, n); d = ctx.synthetic_scalars(1, 0, n)
depths = [int(a) for a in sys.argv[1:]] or [2]
extra = [tuple(int(v) for v in item.split(":")) for item in os.environ.get("DVN_EMULATE_WINDOWS", "").split(",") if item]
c0 = ctx.msm_plan_points(pts, n)[0]
for depth in depths:
    for world, c in [(w, c0) for w in (1, 2, 4, 8)] + extra:
        if c != ctx.msm_plan_points(pts, n)[0]: pts.set_expanded(c)
        pipe = sh.ShardedMsmPipeline(ctx, pts, n, 0, world, FakeDist(), "cpu")
        steps, warm = 40, 5
        for _ in range(depth - 1):
            pipe.submit(d)
        for k in range(warm + steps):
            if k == warm:
                t0 = time.time()
            pipe.submit(d); pipe.collect()
        while pipe.in_flight():
            pipe.collect()
        dt = (time.time() - t0) / steps
        print(f"{depth} steps in flight, window {c}, world {world}: rank-0 step {dt*1e3:.3f} ms -> {n/dt/1e6:.0f} Mscalar/s (ideal {world}x of world 1)", flush=True)
