"""One MSM / one proof shared between the GPUs of a node (BASELINE.json north_star).

Every rank holds the whole device-resident point set and scalar vector and accumulates ITS SHARE of the Pippenger windows:
for an expanded point set (row w = 2^(c w) P, all windows share one bucket set) the share is an equal range of the
windows x n entries (`ShardedMsmPipeline`, dvn_msm_g1_share_*), which balances any world size -- 13 windows over 8 ranks are
1.625 windows each -- and yields ONE partial sum per rank; the older per-window form (`sharded_msm_g1`, dvn_msm_g1_windows)
returns a sum per window.  The exchange is ONE all-gather of 96 bytes (or `ceil(nwin/world) x 128 B`) per rank -- RCCL over
xGMI with backend "nccl", gloo in the CPU tests.  RCCL has no elliptic-curve reduction operator, so the "all-reduce of
partial bucket sums" is all-gather of bytes + a local sum (dvn_msm_g1_sum / dvn_msm_combine_g1).  Latency-bound, never
link-bound.
"""
from __future__ import annotations
import numpy as np


def window_slice(nwin: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, as-even-as-possible slice [lo, hi) of the nwin windows owned by `rank`."""
    return rank * nwin // world, (rank + 1) * nwin // world


def exchange_window_sums(local: np.ndarray, nwin: int, world: int, rank: int, dist=None, device="cpu") -> np.ndarray:
    """all-gather the per-window sums.  local: (hi-lo, 16) uint64 for this rank's slice.  Returns (nwin, 16)."""
    import torch
    lo, hi = window_slice(nwin, world, rank)
    assert local.shape == (hi - lo, 16), (local.shape, lo, hi)
    if dist is None:                                          # one rank without a process group; WITH one the collective runs (rehearsal)
        assert world == 1
        return local
    per = -(-nwin // world)                                   # ranks own unequal counts when nwin % world != 0: pad
    mine = np.zeros((per, 16), dtype=np.uint64)
    mine[: hi - lo] = local
    send = torch.from_numpy(mine.view(np.int64)).to(device)
    recv = torch.empty((world * per, 16), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(recv, send)
    allr = recv.cpu().numpy().view(np.uint64).reshape(world, per, 16)
    out = np.zeros((nwin, 16), dtype=np.uint64)
    for r in range(world):
        l, h = window_slice(nwin, world, r)
        out[l:h] = allr[r, : h - l]
    return out


def sharded_msm_g1(ctx, pts, scalars, n: int, rank: int, world: int, dist=None, device="cuda") -> np.ndarray:
    """One window-sharded G1 MultiExp; every rank returns the same G1Jac image.  Returns (result, device_ms, kernel_ms)."""
    _, nwin, combine_bits = ctx.msm_plan_points(pts, n)
    lo, hi = window_slice(nwin, world, rank)
    sums = ctx.msm_g1_windows(pts, scalars, lo, hi, n=n)
    allsums = exchange_window_sums(sums, nwin, world, rank, dist, device)
    return ctx.msm_combine_g1(allsums, combine_bits)


class ShardedMsmPipeline:
    """G1 MSMs shared by entry count with several steps in flight per rank: the share of step k+1 is queued on the GPU
    (another lane) before the partial sums of step k are exchanged and added, so the all-gather, the host sum and the
    latency-bound tail of one step hide under the accumulate kernel of the next."""

    def __init__(self, ctx, pts, n: int, rank: int, world: int, dist=None, device="cuda"):
        self.ctx, self.pts, self.n, self.rank, self.world, self.dist, self.device = ctx, pts, n, rank, world, dist, device
        self.pending = []

    def submit(self, d_scalars):
        self.pending.append(self.ctx.msm_g1_share_submit(self.pts, d_scalars, self.n, self.rank, self.world))

    def collect(self):
        part = self.ctx.msm_g1_share_collect(self.pending.pop(0))
        return self.ctx.msm_g1_sum(all_gather_rows(part, self.world, self.dist, self.device))

    def in_flight(self) -> int:
        return len(self.pending)


def all_gather_rows(local: np.ndarray, world: int, dist=None, device="cpu") -> np.ndarray:
    """all-gather one fixed-size uint64 row per rank -> (world, len(local))."""
    import torch
    local = np.ascontiguousarray(local, dtype=np.uint64).reshape(-1)
    if dist is None:                                          # one rank without a process group; WITH one the collective runs (rehearsal)
        assert world == 1
        return local.reshape(1, -1)
    send = torch.from_numpy(local.view(np.int64)).to(device)
    recv = torch.empty((world * local.size,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(recv, send)
    return recv.cpu().numpy().view(np.uint64).reshape(world, local.size)


def sharded_groth16_prove(key, witness, r_mont, s_mont, rank: int, world: int, dist=None, device="cuda") -> dict:
    """One statetransition-style proof with its five MSMs shared by entry count across the ranks (BASELINE.json config 5): every
    rank computes its partial results (dvn_groth16_prove_shard), ONE all-gather of 576 bytes per rank, every rank assembles
    the same proof on the host (dvn_groth16_prove_combine)."""
    part = key.prove_shard(witness, rank, world)
    return key.prove_combine(all_gather_rows(part, world, dist, device), r_mont, s_mont)


def sharded_groth16_prove_committed(key, witness, r_mont, s_mont, rank: int, world: int, dist=None, device="cuda") -> dict:
    """The statetransition shape proper (8 public inputs + ONE BSB22 commitment) shared across the ranks: every rank runs the
    commitment hint, the keccak challenge and the witness completion whole (replicated: the commitment bases are small), then
    its entry share of the five MSMs (dvn_groth16_prove_committed_shard); ONE all-gather of 576 bytes per rank; every rank
    assembles the same proof (dvn_groth16_prove_combine) and holds the same Commitment / CommitmentPok."""
    part = key.prove_committed_shard(witness, rank, world)
    proof = key.prove_combine(all_gather_rows(part["partial"], world, dist, device), r_mont, s_mont)
    proof.update({k: part[k] for k in ("Commitment", "CommitmentPok", "challenge")})
    return proof
