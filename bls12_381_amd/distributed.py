"""Multi-GPU sharding of the hot path: one process per GPU, `torch.distributed` over RCCL (backend "nccl").

MSM and multi-Miller-loop shard embarrassingly over their independent terms: each rank reduces its
contiguous slice to ONE group element (144 B G1 / 288 B G2 / 576 B Fp12), the only exchange is an all-gather of
those partials, and every rank folds them in rank order (RCCL has no elliptic-curve / Fp12 reduction operator, so
"reduce" = all-gather + local fold).  Raw buckets are never exchanged.  Batches of independent pairings need no
exchange at all: outputs stay sharded (an optional gather of the N x 576 B results is the caller's choice).

Reference anchors: `Sum for G1Projective` (src/g1.rs:161-171), `MillerLoopResult + MillerLoopResult`
(src/pairings.rs:179-186), `multi_miller_loop` (:554-603), `pairing` (:607-653).

The helpers take the collective as a parameter (`dist`) so that the same code runs over RCCL on GPUs, over gloo in the
CPU tests, and over `LogicalRanks` -- N logical ranks inside one process -- for single-GPU correctness tests.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n items owned by `rank` (sizes differ by at most one)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_partials(partial, world, dist=None, device=None):
    """This rank's partial result -> the (world, len) block of every rank's, on every rank.

    `partial` is either a 1-D np.uint64 array (host limbs: what the synchronous entry points return) or a 1-D int64 torch tensor
    that already lives on the GPU (what the `*_device` entry points write).  A tensor never leaves its device: the rows are
    gathered in place into ONE (world, len) tensor (`all_gather_rows`) which is returned as it is, ready for the device-side
    fold (`blsgpu_g1_sum_device` / `blsgpu_fp12_product_device`).  A numpy partial is staged through `device` (the RCCL backend
    moves device memory only; gloo takes it from the host) and comes back as numpy."""
    is_np = isinstance(partial, np.ndarray)
    if world == 1:
        return partial[None, :].copy() if is_np else partial[None, :].clone()
    import torch
    if dist is None:
        import torch.distributed as dist
    if not is_np:
        gathered = torch.empty((world, partial.shape[0]), dtype=partial.dtype, device=partial.device)
        return all_gather_rows(gathered, partial.contiguous(), dist)
    t = torch.from_numpy(partial.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    gathered = torch.empty((world, t.shape[0]), dtype=t.dtype, device=t.device)
    all_gather_rows(gathered, t, dist)
    return gathered.cpu().numpy().view(np.uint64)


def all_gather_rows(gathered, row, dist=None):
    """The exact tensor plumbing bench.py uses on the RCCL path: `gathered` is ONE (world, words) int64 tensor whose rows
    receive every rank's `row` (same dtype / device) -- the rows are handed to all_gather as views (`unbind`), so the
    result lands in place and can be passed to the device-side fold as one contiguous buffer."""
    if dist is None:
        import torch.distributed as dist
    dist.all_gather(list(gathered.unbind(0)), row)
    return gathered


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


def sharded_pairings(local_pairings, n, world, rank, gather=False, dist=None, device=None):
    """N independent `pairing(p_i, q_i)` (src/pairings.rs:607-653) over `world` ranks: index slices, NO collective on the
    data path -- rank r returns ((lo, hi), Gt values of its slice).  With gather=True the (hi - lo) x 72-limb blocks are
    all-gathered (padded to the largest slice) and every rank returns ((0, n), all n values in index order)."""
    lo, hi = shard_range(n, rank, world)
    mine = np.ascontiguousarray(local_pairings(lo, hi), dtype=np.uint64).reshape(hi - lo, 72)
    if not gather or world == 1:
        return (lo, hi), mine
    import torch
    if dist is None:
        import torch.distributed as dist
    width = -(-n // world)
    pad = np.zeros((width, 72), dtype=np.uint64)
    pad[:hi - lo] = mine
    t = torch.from_numpy(pad.view(np.int64))
    if device is not None:
        t = t.to(device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    rows = []
    for r in range(world):
        l, h = shard_range(n, r, world)
        rows.append(out[r].cpu().numpy().view(np.uint64)[:h - l])
    return (0, n), np.concatenate(rows, axis=0)


class LogicalRanks:
    """`world` logical ranks inside ONE process (all on one GPU): a stand-in for the process group in single-GPU
    correctness tests of the rank logic (SURVEY.md 8e caveat).  Usage: run the per-rank function once per rank with
    `dist = lr.view(rank)`; collectives rendezvous through the shared store, so the ranks must be driven in lock-step
    phases: every rank calls `contribute`, then every rank calls `collect`."""

    def __init__(self, world):
        self.world = world
        self.slots = {}

    def contribute(self, key, rank, value):
        self.slots.setdefault(key, [None] * self.world)[rank] = np.array(value, copy=True)

    def collect(self, key):
        parts = self.slots[key]
        assert all(p is not None for p in parts), "LogicalRanks: a rank has not contributed to " + str(key)
        return np.stack(parts)
