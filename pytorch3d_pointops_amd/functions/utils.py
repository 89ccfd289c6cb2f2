"""masked_gather, wmean, get_point_covariances -- API of the reference's functions/utils.py.

reference: pytorch3d_pointops/functions/utils.py:20-65 (masked_gather), :68-108
(wmean), :111-153 (get_point_covariances).
"""
from typing import Optional, Tuple, Union

import torch

from .. import _C


def masked_gather(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Gather ``points`` (N,P,D) at ``idx`` (N,K) or (N,P',K) where -1 marks padding.

    Same contract as the reference (functions/utils.py:20-65): padded entries give
    0.0; differentiable w.r.t. ``points``.  Runs the fused HIP gather kernel
    (idx < 0 -> 0) instead of clone + masked index + torch.gather + masked fill.
    """
    from .knn import _gather_neighbors

    if len(idx) != len(points):
        raise ValueError("points and idx must have the same batch dimension")
    if idx.ndim not in (2, 3):
        raise ValueError("idx format is not supported %s" % repr(idx.shape))
    if points.dtype != torch.float32:
        return _masked_gather_torch(points, idx)
    table = idx if idx.ndim == 3 else idx[:, :, None]
    if torch.compiler.is_compiling():
        out = torch.ops.pointops_amd.gather_neighbors(points, table.contiguous(), None)
    elif not (torch.is_grad_enabled() and points.requires_grad):
        out = _C.gather_neighbors(points, table, None)  # nothing to differentiate: no autograd node
    else:
        out = _gather_neighbors.apply(points, table, None)
    return out if idx.ndim == 3 else out[:, :, 0, :]


def _masked_gather_torch(points, idx):
    N, P, D = points.shape
    mask = idx.eq(-1)
    safe = idx.masked_fill(mask, 0)
    if idx.ndim == 3:
        K = idx.shape[2]
        out = points[:, :, None, :].expand(-1, -1, K, -1).gather(1, safe[..., None].expand(-1, -1, -1, D))
    else:
        out = points.gather(1, safe[..., None].expand(-1, -1, D))
    return out.masked_fill(mask[..., None], 0.0)


def wmean(
    x: torch.Tensor,
    weight: Optional[torch.Tensor] = None,
    dim: Union[int, Tuple[int]] = -2,
    keepdim: bool = True,
    eps: float = 1e-9,
) -> torch.Tensor:
    """(Weighted) mean over ``dim``; reference: functions/utils.py:68-108."""
    args = {"dim": dim, "keepdim": keepdim}
    if weight is None:
        return x.mean(**args)
    if any(xd != wd and xd != 1 and wd != 1 for xd, wd in zip(x.shape[-2::-1], weight.shape[::-1])):
        raise ValueError("wmean: weights are not compatible with the tensor")
    return (x * weight[..., None]).sum(**args) / weight[..., None].sum(**args).clamp(eps)


def get_point_covariances(
    points_padded: torch.Tensor,
    num_points_per_cloud: torch.Tensor,
    neighborhood_size: int,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-point covariance of the K nearest neighbours; reference: functions/utils.py:111-153."""
    from .knn import knn_points

    k_nearest_neighbors = knn_points(
        points_padded,
        points_padded,
        lengths1=num_points_per_cloud,
        lengths2=num_points_per_cloud,
        K=neighborhood_size,
        return_nn=True,
    ).knn
    if (k_nearest_neighbors.is_cuda and k_nearest_neighbors.dtype == torch.float32
            and 1 <= k_nearest_neighbors.shape[3] <= _C.POINT_COVARIANCES_MAX_D and k_nearest_neighbors.shape[2] >= 1):
        # fused: no (N,P,K,D,D) tensor of outer products, closed-form backward (csrc/covariance.hip)
        if torch.compiler.is_compiling():
            return torch.ops.pointops_amd.point_covariances(k_nearest_neighbors.contiguous()), k_nearest_neighbors
        return _point_covariances.apply(k_nearest_neighbors), k_nearest_neighbors
    pt_mean = k_nearest_neighbors.mean(2, keepdim=True)
    central_diff = k_nearest_neighbors - pt_mean
    per_pt_cov = central_diff.unsqueeze(4) * central_diff.unsqueeze(3)
    covariances = per_pt_cov.mean(2)
    return covariances, k_nearest_neighbors


class _point_covariances(torch.autograd.Function):
    """cov[a][b] = mean_k (x_k[a] - m[a])(x_k[b] - m[b]); d/dx_k = (G + G^T)(x_k - m) / K."""

    @staticmethod
    def forward(ctx, knn):
        knn = knn.contiguous()
        ctx.save_for_backward(knn)
        return _C.point_covariances(knn)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_cov):
        (knn,) = ctx.saved_tensors
        return _C.point_covariances_backward(knn, grad_cov.contiguous().float())
