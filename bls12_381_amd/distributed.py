"""Multi-GPU sharding of the hot path: one process per GPU, `torch.distributed` over RCCL (backend "nccl").

MSM and multi-Miller-loop shard embarrassingly over their independent terms: each rank reduces its
contiguous slice to ONE group element (144 B G1 / 288 B G2 / 576 B Fp12), the only exchange is an all-gather of
those partials, and every rank folds them in rank order (RCCL has no elliptic-curve / Fp12 reduction operator, so
"reduce" = all-gather + local fold).  Raw buckets are never exchanged.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n items owned by `rank` (sizes differ by at most one)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_partials(partial, world, dist=None, device=None):
    """partial: 1-D np.uint64 array (wire limbs of this rank's partial result) -> (world, len) array on every rank."""
    if world == 1:
        return partial[None, :].copy()
    import torch
    if dist is None:
        import torch.distributed as dist
    t = torch.from_numpy(partial.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def sharded_msm(local_msm, fold, scalars, world, rank, dist=None, device=None):
    """Run one MSM over `scalars` (n x 32 uint8) sharded over `world` ranks.

    local_msm(lo, hi, scalars[lo:hi]) -> projective wire limbs of sum_{i in [lo,hi)} s_i * P_i   (this rank's GPU)
    fold(parts)                        -> projective wire limbs of the sum of the rows of `parts`  (blsgpu_g1_sum)
    Every rank returns the same full result."""
    lo, hi = shard_range(len(scalars), rank, world)
    part = np.ascontiguousarray(local_msm(lo, hi, scalars[lo:hi]), dtype=np.uint64)
    parts = all_gather_partials(part, world, dist, device)
    return fold(parts)


def sharded_product(local_product, fold, n, world, rank, dist=None, device=None):
    """Same pattern for multi_miller_loop: rank-local Fp12 product of its terms, all-gather, Fp12 fold, then ONE
    final exponentiation by the caller."""
    lo, hi = shard_range(n, rank, world)
    part = np.ascontiguousarray(local_product(lo, hi), dtype=np.uint64)
    return fold(all_gather_partials(part, world, dist, device))
