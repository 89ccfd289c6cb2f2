"""sample_farthest_points -- same API as the reference's functions/sample_farthest_points.py.

reference: pytorch3d_pointops/functions/sample_farthest_points.py:18-96
(sample_farthest_points), :99-197 (sample_farthest_points_naive).
"""
from random import randint
from typing import List, Optional, Tuple, Union

import torch

from .. import _C
from ._common import as_f32, batch_vector, full_lengths, lengths_max
from .utils import masked_gather


def _per_cloud(points, lengths, K, too_large_msg):
    """(lengths, K) as (N,) int64 device vectors, checked the way the reference checks them (:55-70, :131-146)."""
    n, p, _ = points.shape
    if lengths is None:
        lengths = full_lengths(n, p, points.device)
    else:
        if lengths.shape != (n,):
            raise ValueError("points and lengths must have same batch dimension.")
        if lengths_max(lengths) > p:
            raise ValueError(too_large_msg)
        lengths = batch_vector(lengths, n, points.device, "lengths", "points and lengths must have same batch dimension.")
    return lengths, batch_vector(K, n, points.device, "K", "K and points must have the same batch dimension")


def sample_farthest_points(
    points: torch.Tensor,
    lengths: Optional[torch.Tensor] = None,
    K: Union[int, List, torch.Tensor] = 50,
    random_start_point: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Iterative farthest point sampling; same contract as the reference
    (functions/sample_farthest_points.py:18-96).

    Returns ``(selected_points (N, max K, D), selected_indices (N, max K))``; indices
    are -1 and points 0.0 beyond ``min(lengths[n], K[n])``.  Indices are computed
    without autograd; the points come from a differentiable gather.
    """
    known_max = K if isinstance(K, int) else max(K) if isinstance(K, (list, tuple)) and len(K) else None
    lengths, K = _per_cloud(points, lengths, K, "A value in lengths was too large.")
    points = as_f32(points)
    start_idxs = torch.zeros_like(lengths)
    if random_start_point:  # one torch.randint draw per cloud, the reference's RNG consumption (:86-89)
        for n in range(points.shape[0]):
            start_idxs[n] = torch.randint(high=lengths[n], size=(1,)).item()

    with torch.no_grad():
        if torch.compiler.is_compiling():
            idx = torch.ops.pointops_amd.sample_farthest_points(points, lengths, K, start_idxs)
        else:
            idx = _C.sample_farthest_points(points, lengths, K, start_idxs, max_K=known_max)
    sampled_points = masked_gather(points, idx)
    return sampled_points, idx


def sample_farthest_points_naive(
    points: torch.Tensor,
    lengths: Optional[torch.Tensor] = None,
    K: Union[int, List, torch.Tensor] = 50,
    random_start_point: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pure-torch per-cloud FPS (any device); same Args/Returns as
    sample_farthest_points.  reference: functions/sample_farthest_points.py:99-197.
    Kept as the readable cross-check the reference's examples use
    (examples/fps_on_pointclouds.py:122-155); it is not the product path.
    """
    lengths, K = _per_cloud(points, lengths, K, "Invalid lengths.")
    N, P, D = points.shape
    device = points.device
    max_K = int(torch.max(K))
    rows = []
    for n in range(N):
        row = torch.full((max_K,), -1, dtype=torch.int64, device=device)
        len_n = int(lengths[n])
        closest = points.new_full((len_n,), float("inf"), dtype=torch.float32)
        sel = randint(0, len_n - 1) if random_start_point else 0
        row[0] = sel
        k_n = min(len_n, int(K[n]))
        cloud = points[n, :len_n, :]
        for i in range(1, k_n):
            delta = points[n, sel, :] - cloud
            closest = torch.min((delta ** 2).sum(-1), closest)
            sel = torch.argmax(closest)
            row[i] = sel
        rows.append(row)
    all_idx = torch.stack(rows, dim=0)
    if points.is_cuda and points.dtype == torch.float32:
        return masked_gather(points, all_idx), all_idx
    from .utils import _masked_gather_torch

    return _masked_gather_torch(points, all_idx), all_idx
