"""How the CPU port of the MSM scales with threads on this host (context for bench.py's cpu_baseline)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu as orc
orc.build()
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp threads", orc.num_threads())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup cpu.max: n/a", e)
n = 1 << 20
pts = orc.g1_synthetic_points(1, n); sc = orc.scalars(1, 0, n)
for th in (1, 8, 32, 64, 128):
    if th > os.cpu_count(): break
    t = time.time(); orc.msm_g1(pts, sc, th); dt = time.time() - t
    print(f"threads {th:3d}: {dt:.3f} s -> {n/dt/1e6:.2f} Mscalar/s", flush=True)
