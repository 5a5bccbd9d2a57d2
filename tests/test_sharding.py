"""N > 1 path on CPU: world_size-2 gloo run of the window-sharded MSM exchange + combine.  The per-window
sums come from the oracle here (no GPU in this container); the exchange and the Horner combine are the product
code (davinci-node_amd/sharding.py, dvn_msm_combine_g1)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, c, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    from oracle import cpu as orc
    from davinci_node_amd import sharding, binding
    import ctypes as C
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pts = orc.g1_synthetic_points(0xDA71C1, n)
    sc = orc.scalars(1, 0, n)
    nwin = 254 // c + 1
    lo, hi = sharding.window_slice(nwin, world, rank)
    local = orc.msm_g1_window_sums(pts, sc, c)[lo:hi]                 # stands in for dvn_msm_g1_windows on this rank
    allsums = sharding.exchange_window_sums(np.ascontiguousarray(local), nwin, world, rank, dist, "cpu")
    L = pkg.load_library()
    out = np.zeros(12, dtype=np.uint64)
    rc = L.dvn_msm_combine_g1(None, C.c_void_p(allsums.ctypes.data), nwin, c, C.c_void_p(out.ctypes.data))
    assert rc == 0
    got = orc.g1_jac_to_affine(out)
    want = orc.msm_g1(pts, sc)
    # the fixed-size row all-gather of the sharded Groth16 prover (72 uint64 of partial results per rank)
    rows = sharding.all_gather_rows(np.arange(72, dtype=np.uint64) + 1000 * rank, world, dist, "cpu")
    rows_ok = rows.shape == (world, 72) and all((rows[r] == np.arange(72, dtype=np.uint64) + 1000 * r).all() for r in range(world))
    # the shared-by-entry-count form (dvn_msm_g1_share_* on a GPU): ONE partial sum per rank, all-gather of 96 bytes, host sum
    # (dvn_msm_g1_sum).  Here a rank's share is the oracle's MSM over its half of the points.
    lo_p, hi_p = rank * n // world, (rank + 1) * n // world
    one = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], dtype=np.uint64)     # R mod p
    part = np.concatenate([orc.msm_g1(pts[lo_p:hi_p], sc[lo_p:hi_p]), one])
    gathered = sharding.all_gather_rows(part, world, dist, "cpu")
    tot = np.zeros(12, dtype=np.uint64)
    assert L.dvn_msm_g1_sum(C.c_void_p(gathered.ctypes.data), world, C.c_void_p(tot.ctypes.data)) == 0
    share_ok = bool((orc.g1_jac_to_affine(tot) == want).all())
    q.put((rank, bool((got == want).all()) and rows_ok and share_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("c", [16, 13])          # 16 windows (even split) and 20 windows
def test_window_sharded_msm_gloo_world2(c):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + c
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 3000, c, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_window_slices_cover_all_windows():
    from __graft_entry__ import load_package
    load_package()
    from davinci_node_amd import sharding
    for nwin in (16, 17, 20, 26, 43):
        for world in (1, 2, 3, 4, 8):
            cuts = [sharding.window_slice(nwin, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nwin
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))


def test_exchange_with_uneven_slices_world3():
    """Padding logic with nwin % world != 0, single process (simulated gather)."""
    from __graft_entry__ import load_package
    load_package()
    from davinci_node_amd import sharding

    class FakeDist:
        def __init__(self, parts): self.parts = parts
        def all_gather_into_tensor(self, recv, send):
            import torch
            recv.copy_(torch.cat(self.parts))

    import torch
    nwin, world = 20, 3
    full = np.arange(nwin * 16, dtype=np.uint64).reshape(nwin, 16)
    per = -(-nwin // world)
    padded = []
    for r in range(world):
        lo, hi = sharding.window_slice(nwin, world, r)
        m = np.zeros((per, 16), dtype=np.uint64); m[: hi - lo] = full[lo:hi]
        padded.append(torch.from_numpy(m.view(np.int64)))
    for r in range(world):
        lo, hi = sharding.window_slice(nwin, world, r)
        out = sharding.exchange_window_sums(np.ascontiguousarray(full[lo:hi]), nwin, world, r, FakeDist(padded), "cpu")
        assert (out == full).all()


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The real `bench.py --gpus 2` path (one process per rank, torch.distributed rendezvous, the MSM shared by entry count,
    all-gather, host sum, result check) with both ranks on the one GPU of the box and gloo as the exchange backend."""
    import json
    import subprocess
    env = dict(os.environ, DVN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--log-n", "20", "--no-extras", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["result_checked"] is True and line["value"] > 0
    assert line["roofline"]["pairs_per_launch"] == (1 << 20) // 2        # per-rank algorithmic bytes, not the whole MSM's


@pytest.mark.gpu
def test_bench_two_ranks_extras_prove_the_committed_statetransition_shape():
    """`bench.py --gpus 2` WITH its multi-GPU extras (two ranks on the one GPU, gloo): the sharded leg proves the 8-public +
    1-commitment key (BASELINE config 5's shape), all ranks hold the same proof, it equals the single-GPU proof, and the
    process exits 0."""
    import json
    import subprocess
    env = dict(os.environ, DVN_BENCH_BACKEND="gloo", DVN_BENCH_EXTRAS_LIMIT_S="250")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--log-n", "20", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=290, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    ex = line["extras"]
    assert "error" not in ex, ex
    sh = ex["groth16_sharded_2^20"]
    assert sh["ranks"] == 2 and sh["all_ranks_same_proof"] is True and sh["equals_single_gpu_proof"] is True and "1 commitment" in sh["shape"]
    assert ex["groth16_replicas_2^20"]["post_solver_proofs_per_s"] > 0


_RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
from davinci_node_amd import sharding
from oracle import cpu as orc
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()
except Exception as e:                                   # the backend itself is unusable on this machine: nothing of ours has run yet
    print("RCCL_UNAVAILABLE", repr(e)[:300], flush=True); raise SystemExit(0)
ctx = pkg.Context(0)
n = 1 << 16
pts = ctx.synthetic_g1(5, n); d = ctx.synthetic_scalars(3, 0, n)
want = orc.g1_jac_to_affine(ctx.msm_g1(pts, d, n=n))
pipe = sharding.ShardedMsmPipeline(ctx, pts, n, 0, 1, dist, "cuda")       # the N > 1 code of bench.py, one rank
pipe.submit(d); pipe.submit(d)
a = pipe.collect(); b = pipe.collect()
assert (orc.g1_jac_to_affine(a) == want).all() and (orc.g1_jac_to_affine(b) == want).all()
ctx.synchronize(); torch.cuda.synchronize(); dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 1.5
cs = pkg.R1CS(7, 3000, 9, n_committed=100); key = pkg.Groth16Key(ctx, cs, 3); dw = ctx.to_device(cs.witness())
rs = np.zeros((2, 4), dtype=np.uint64); rs[0, 0], rs[1, 0] = 5, 7
proof = sharding.sharded_groth16_prove_committed(key, dw, rs[0], rs[1], 0, 1, dist, "cuda")
single = key.prove_committed(dw, rs[0], rs[1])
assert all((single[k] == proof[k]).all() for k in ("Ar", "Bs", "Krs", "Commitment", "CommitmentPok"))
same = torch.from_numpy(np.concatenate([proof["Ar"], proof["Bs"], proof["Krs"]]).view(np.int64)).to("cuda")
ref = same.clone(); dist.broadcast(ref, 0)
agree = torch.tensor([int(bool((same == ref).all()))], device="cuda"); dist.all_reduce(agree, op=dist.ReduceOp.MIN); assert int(agree.item()) == 1
rows = sharding.exchange_window_sums(np.arange(16 * 16, dtype=np.uint64).reshape(16, 16), 16, 1, 0, dist, "cuda")
assert (rows == np.arange(16 * 16, dtype=np.uint64).reshape(16, 16)).all()
dist.barrier(); dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", flush=True)
"""


@pytest.mark.gpu
def test_rccl_collectives_of_the_sharded_path_with_one_rank():
    """Every torch.distributed call of the N > 1 path (bench.py, sharding.py) on the real backend, "nccl" = RCCL, with device
    tensors: process-group init with a device id, barrier, all-gather of int64 views, all-reduce MAX / MIN, broadcast.  RCCL
    refuses two ranks on one device, so the rehearsal on a one-GPU box is a world of ONE rank; the two-rank runs above cover
    the exchange logic over gloo.  Own process: the process group must not outlive the test."""
    import subprocess
    code = _RCCL_ONE_RANK.format(root=ROOT, port=29400 + os.getpid() % 90)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200, cwd=ROOT)
    if "RCCL_UNAVAILABLE" in out.stdout:
        pytest.skip("torch's nccl backend does not initialise on this machine: " + out.stdout[-300:])
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.gpu
def test_bench_distributed_path_on_rccl_with_one_rank():
    """`bench.py`'s own N > 1 code (process group on "nccl" = RCCL with a device id, the shared-MSM pipeline with its
    all-gather on device tensors, max-over-ranks timing, the sharded committed proof and the replica leg of the extras,
    the final barrier) launched by torch.distributed.run with ONE rank: DVN_BENCH_REHEARSE_DIST=1 makes a world of one take
    that path.  What a one-GPU box cannot show is an exchange between two devices."""
    import json
    import subprocess
    env = dict(os.environ, DVN_BENCH_REHEARSE_DIST="1", DVN_BENCH_EXTRAS_LIMIT_S="250")
    env.pop("DVN_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--log-n", "20", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=290, cwd=ROOT)
    if out.returncode != 0 and "init_process_group" in out.stderr and '"metric"' not in out.stdout:
        pytest.skip("torch's nccl backend does not initialise on this machine: " + out.stderr[-300:])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["result_checked"] is True and line["value"] > 0
    assert "shared by entry count" in line["config"]["parallelism"]
    ex = line["extras"]
    assert "error" not in ex, ex
    sh = ex["groth16_sharded_2^20"]
    assert sh["ranks"] == 1 and sh["all_ranks_same_proof"] is True and sh["equals_single_gpu_proof"] is True


_COMM_ONE_RANK = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cpu as orc
ctx = pkg.Context(0)
try:
    uid = pkg.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = pkg.Comm(ctx, uid, 0, 1)
except pkg.DvnError as e:                                # librccl missing or unable to initialise here: the library says so, nothing is wrong with it
    print("RCCL_UNAVAILABLE", str(e)[:300], flush=True); raise SystemExit(0)
n = 1 << 16
pts = ctx.synthetic_g1(5, n); d = ctx.synthetic_scalars(3, 0, n)
want = orc.g1_jac_to_affine(ctx.msm_g1(pts, d, n=n))
part = ctx.msm_g1_share_collect(ctx.msm_g1_share_submit(pts, d, n, 0, 1))
rows = comm.all_gather(part)                                   # the exchange of a shared MSM: 12 words per rank
assert rows.shape == (1, 12) and (rows[0] == part).all()
assert (orc.g1_jac_to_affine(ctx.msm_g1_sum(rows)) == want).all()
cs = pkg.R1CS(7, 3000, 9); key = pkg.Groth16Key(ctx, cs, 3); dw = ctx.to_device(cs.witness())
rs = np.zeros((2, 4), dtype=np.uint64); rs[0, 0], rs[1, 0] = 5, 7
single = key.prove(dw, rs[0], rs[1])
proof = key.prove_combine(comm.all_gather(key.prove_shard(dw, 0, 1)), rs[0], rs[1])      # ... and of a shared proof: 72 words
assert all((single[k] == proof[k]).all() for k in single)
big = np.arange(4096, dtype=np.uint64)
assert (comm.all_gather(big)[0] == big).all()
for bad in (np.zeros(4097, dtype=np.uint64), np.zeros(0, dtype=np.uint64)):
    try:
        comm.all_gather(bad); raise SystemExit("oversized / empty all-gather was accepted")
    except pkg.DvnError as e:
        assert "between 1 and 4096" in str(e)
try:
    pkg.Comm(ctx, uid, 3, 2); raise SystemExit("rank outside the world was accepted")
except pkg.DvnError:
    pass
comm.free()
print("COMM_ONE_RANK_OK", flush=True)
"""


@pytest.mark.gpu
def test_library_communicator_on_rccl_with_one_rank():
    """dvn_comm_* (the exchange primitive a one-process-per-GPU Go deployment binds, RCCL opened by the library itself, no torch):
    id, init, all-gather of an MSM share and of a proof share, limits, free -- with a world of one rank, which is what a
    one-GPU box can run (RCCL refuses two ranks on one device).  Own process: RCCL keeps threads and state."""
    import subprocess
    out = subprocess.run([sys.executable, "-c", _COMM_ONE_RANK.format(root=ROOT)], capture_output=True, text=True, timeout=200, cwd=ROOT)
    if "RCCL_UNAVAILABLE" in out.stdout:
        pytest.skip("RCCL does not initialise on this machine: " + out.stdout[-300:])
    assert out.returncode == 0 and "COMM_ONE_RANK_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_library_communicator_rejects_bad_arguments_without_a_device():
    import ctypes as C
    from __graft_entry__ import load_package
    L = load_package().load_library()
    assert L.dvn_comm_unique_id(None) != 0
    assert L.dvn_comm_free(None, None) == 0                    # freeing nothing is not an error
    assert L.dvn_comm_init(None, None, 0, 1, None) != 0 and L.dvn_comm_all_gather(None, None, None, 1, None) != 0


def test_bench_exit_status_reports_failed_extras():
    """Source-level contract of bench.py's N > 1 tail: both failure paths (watchdog, exception in the extras) leave with a
    non-zero status after printing the line, never 0."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tail = src[src.index("def bail():"):]
    assert tail.count("os._exit(3)") == 2 and "os._exit(0)" not in src
