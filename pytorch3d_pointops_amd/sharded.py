"""Batch sharding of the neighbour path over the GPUs of one node.

Clouds are independent units in every op (reference kernels index by batch first:
csrc/knn/knn_cpu.cpp:35, ball_query_cpu.cpp:35, sample_farthest_points_cpu.cpp:45), so
the batch is partitioned over ranks -- one process per GPU, `torch.distributed` backend
"nccl" (= RCCL over xGMI on ROCm).  KNN / ball-query / FPS outputs stay sharded: there is
NO data-path collective.  The only exchange the path has is chamfer's batch reduction:
one all_gather of the per-cloud loss vectors (4*B/G bytes per rank and loss term --
latency-bound, so a single RCCL all_gather, no custom ring), after which every rank
reduces the full vector in fixed cloud order (deterministic, identical on all ranks).
The reference has no counterpart (no distributed code at all).
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_clouds: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal split of `num_clouds` over `world_size` ranks."""
    base, rem = divmod(num_clouds, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


def balanced_assignment(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time bin packing of clouds by cost (for ragged chamfer the
    cost of cloud n is len1[n]*len2[n]); returns the cloud ids of every rank, each sorted."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    bins: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        bins[r].append(i)
        loads[r] += float(costs[i])
    return [sorted(b) for b in bins]


class _AllGatherVec(torch.autograd.Function):
    """all_gather of equally sized 1-D tensors; backward returns this rank's slice of the
    upstream gradient (every rank back-propagates the same full-batch loss, and clouds are
    not shared between ranks, so no reduce is needed)."""

    @staticmethod
    def forward(ctx, local: torch.Tensor, group) -> torch.Tensor:
        world = dist.get_world_size(group)
        ctx.rank = dist.get_rank(group)
        ctx.n = local.shape[0]
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, grad):
        return grad[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n].contiguous(), None


def all_gather_losses(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """Gather per-cloud loss vectors of (possibly different) lengths `counts[r]` from all
    ranks into one (sum(counts),) vector in rank order.  Differentiable."""
    world = dist.get_world_size(group)
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank(group)]
    m = max(counts) if counts else 0
    padded = torch.zeros((m,), dtype=local.dtype, device=local.device)
    padded = torch.cat([local, padded[local.shape[0]:]]) if local.shape[0] < m else local
    full = _AllGatherVec.apply(padded, group)
    if all(c == m for c in counts):
        return full
    return torch.cat([full[r * m:r * m + counts[r]] for r in range(world)])


def plan_shards(num_clouds: int, world_size: int, costs: Optional[Sequence[float]] = None) -> List[List[int]]:
    """Global cloud ids owned by every rank.  Without `costs`: the contiguous `shard_bounds` split.
    With `costs` (ragged chamfer: len1[n]*len2[n]): cost-balanced placement (`balanced_assignment`),
    so a batch of 20k..200k-point clouds does not leave ranks idle behind the one that drew the big clouds."""
    if costs is None:
        return [list(range(s, e)) for s, e in shard_bounds(num_clouds, world_size)]
    assert len(costs) == num_clouds
    return balanced_assignment(costs, world_size)


def sharded_chamfer_distance(
    x_local, y_local,
    num_clouds_total: int,
    *,
    weights_local: Optional[torch.Tensor] = None,
    weights_sum_total: Optional[float] = None,
    batch_reduction: Optional[str] = "mean",
    group=None,
    local_fn: Optional[Callable] = None,
    cloud_counts: Optional[Sequence[int]] = None,
    assignment: Optional[Sequence[Sequence[int]]] = None,
    **chamfer_kwargs,
):
    """chamfer_distance over a batch sharded across ranks.

    Placement: by default the contiguous `shard_bounds` split; `cloud_counts` gives explicit per-rank
    counts of a contiguous split; `assignment` (from `plan_shards(..., costs)`) gives the global cloud
    ids of every rank -- rank r's local tensors then hold clouds assignment[r] in that order, and the
    gathered vector is un-permuted into global cloud order.

    Each rank computes the per-cloud losses of ITS clouds with the single-GPU path
    (`batch_reduction=None`), the per-rank vectors are all-gathered, and the reference's
    batch reduction (functions/chamfer.py:192-214) is applied to the full (B,) vector in global cloud
    order, so the value equals the single-process result on the whole batch whatever the placement.
    Returns (loss, loss_features) like chamfer_distance; per-cloud vectors when
    batch_reduction is None (full batch, global cloud order).
    `local_fn` (tests only) replaces the local per-cloud computation.
    """
    if batch_reduction not in (None, "mean", "sum"):
        raise ValueError('batch_reduction must be one of ["mean", "sum"] or None')
    if chamfer_kwargs.get("point_reduction", "mean") is None:
        raise ValueError("sharded_chamfer_distance needs a point_reduction (per-cloud scalars)")
    world = dist.get_world_size(group)
    if assignment is not None:
        if len(assignment) != world or sorted(i for a in assignment for i in a) != list(range(num_clouds_total)):
            raise ValueError("assignment must list every cloud id exactly once over world_size ranks")
        counts = [len(a) for a in assignment]
    elif cloud_counts is not None:
        counts = [int(c) for c in cloud_counts]
        if len(counts) != world or sum(counts) != num_clouds_total:
            raise ValueError("cloud_counts must have world_size entries that sum to num_clouds_total")
    else:
        counts = [e - s for s, e in shard_bounds(num_clouds_total, world)]
    if local_fn is None:
        from .functions.chamfer import chamfer_distance

        def local_fn(x, y, **kw):
            return chamfer_distance(x, y, batch_reduction=None, **kw)

    unpermute = None
    if assignment is not None:
        order = [i for a in assignment for i in a]  # gathered position -> global cloud id
        if order != list(range(num_clouds_total)):
            inv = [0] * num_clouds_total
            for pos, gid in enumerate(order):
                inv[gid] = pos

            def unpermute(v):
                return v[torch.tensor(inv, dtype=torch.int64, device=v.device)]

    def gather(v):
        full = all_gather_losses(v, counts, group)
        return unpermute(full) if unpermute is not None else full

    loss_l, feat_l = local_fn(x_local, y_local, weights=weights_local, **chamfer_kwargs)
    loss = gather(loss_l)
    feats: Optional[Dict[str, torch.Tensor]] = None
    if feat_l is not None:
        feats = {k: gather(v) for k, v in sorted(feat_l.items())}
    if batch_reduction is None:
        return loss, feats
    loss = loss.sum()
    if feats is not None:
        feats = {k: v.sum() for k, v in feats.items()}
    if batch_reduction == "mean":
        if weights_local is None:
            div = max(num_clouds_total, 1)
        else:
            if weights_sum_total is None:
                t = weights_local.sum().detach().clone()
                dist.all_reduce(t, group=group)
                weights_sum_total = float(t)
            div = 1.0 if weights_sum_total == 0.0 else weights_sum_total
        loss = loss / div
        if feats is not None:
            feats = {k: v / div for k, v in feats.items()}
    return loss, feats
