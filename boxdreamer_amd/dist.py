"""Multi-GPU eval sweep: one process per GPU, the batch sharded across ranks, weights replicated.

Samples (a query with its references) are independent units, so the data path needs
no collective.  The only exchange is the all-gather of predicted corners -- (B_local, 8, 2) fp32,
<= 4 KB per rank -- which replaces the reference's pickle + gloo `comm.gather` and `dist.barrier`
(/root/reference/src/utils/comm.py:179-219, src/lightning/BoxDreamer_lightning_model.py:248-289).
On the GPU box the backend is "nccl" (= RCCL over xGMI); the CPU tests use "gloo".
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n samples for `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(data: dict, rank: int, world: int) -> dict:
    """Slice every batched tensor of a BoxDreamer batch dict along dim 0."""
    n = data["images"].shape[0]
    lo, hi = shard_range(n, rank, world)
    return {k: (v[lo:hi] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v) for k, v in data.items()}


def gather_corners(kp: torch.Tensor, world: int | None = None, group=None) -> torch.Tensor:
    """All-gather equal-sized per-rank corner tensors (B_local, ...) -> (world * B_local, ...), rank-major.
    One latency-bound collective on a preallocated buffer; no pickling."""
    if world is None:
        world = dist.get_world_size(group)
    kp = kp.contiguous()
    out = torch.empty((world * kp.shape[0], *kp.shape[1:]), dtype=kp.dtype, device=kp.device)
    dist.all_gather_into_tensor(out, kp, group=group)
    return out


def gather_corners_ragged(kp: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Same for ragged shards produced by shard_range: pad to the largest shard, gather once, trim."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cap = -(-n_total // world)
    pad = torch.zeros((cap, *kp.shape[1:]), dtype=kp.dtype, device=kp.device)
    pad[: kp.shape[0]] = kp
    allk = gather_corners(pad, world, group).reshape(world, cap, *kp.shape[1:])
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(allk[r, : hi - lo])
    return torch.cat(parts, 0)
