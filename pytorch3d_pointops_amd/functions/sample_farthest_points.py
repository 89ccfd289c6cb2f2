"""sample_farthest_points -- same API as the reference's functions/sample_farthest_points.py.

reference: pytorch3d_pointops/functions/sample_farthest_points.py:18-96
(sample_farthest_points), :99-197 (sample_farthest_points_naive).
"""
from random import randint
from typing import List, Optional, Tuple, Union

import torch

from .. import _C
from .utils import masked_gather


def _prepare(points, lengths, K, too_large_msg):
    N, P, D = points.shape
    device = points.device
    if lengths is None:
        lengths = torch.full((N,), P, dtype=torch.int64, device=device)
    else:
        if lengths.shape != (N,):
            raise ValueError("points and lengths must have same batch dimension.")
        if lengths.max() > P:
            raise ValueError(too_large_msg)
    if isinstance(K, int):
        K = torch.full((N,), K, dtype=torch.int64, device=device)
    elif isinstance(K, list):
        K = torch.tensor(K, dtype=torch.int64, device=device)
    if K.shape[0] != N:
        raise ValueError("K and points must have the same batch dimension")
    return lengths, K


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
    lengths, K = _prepare(points, lengths, K, "A value in lengths was too large.")
    N = points.shape[0]

    if not (points.dtype == torch.float32):
        points = points.to(torch.float32)
    if not (lengths.dtype == torch.int64):
        lengths = lengths.to(torch.int64)
    if not (K.dtype == torch.int64):
        K = K.to(torch.int64)
    K = K.to(points.device)

    start_idxs = torch.zeros_like(lengths)
    if random_start_point:
        # same RNG consumption as the reference (:86-89): one torch.randint per cloud
        for n in range(N):
            start_idxs[n] = torch.randint(high=lengths[n], size=(1,)).item()

    with torch.no_grad():
        if torch.compiler.is_compiling():
            idx = torch.ops.pointops_amd.sample_farthest_points(points, lengths, K, start_idxs)
        else:
            idx = _C.sample_farthest_points(points, lengths, K, start_idxs)
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
    lengths, K = _prepare(points, lengths, K, "Invalid lengths.")
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
