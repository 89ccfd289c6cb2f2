"""ball_query -- same API as the reference's functions/ball_query.py.

reference: pytorch3d_pointops/functions/ball_query.py:20-52 (_ball_query), :55-142.
"""
from typing import Union

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from .knn import _KNN, _full_lengths
from .utils import masked_gather


class _ball_query(Function):
    @staticmethod
    def forward(ctx, p1, p2, lengths1, lengths2, K, radius):
        idx, dists = _C.ball_query(p1, p2, lengths1, lengths2, K, radius)
        ctx.save_for_backward(p1, p2, lengths1, lengths2, idx)
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_dists, grad_idx):
        p1, p2, lengths1, lengths2, idx = ctx.saved_tensors
        if not (grad_dists.dtype == torch.float32):
            grad_dists = grad_dists.float()
        if not (p1.dtype == torch.float32):
            p1 = p1.float()
        if not (p2.dtype == torch.float32):
            p2 = p2.float()
        # the KNN backward with norm=2; idx == -1 entries are skipped by the kernel
        grad_p1, grad_p2 = _C.knn_points_backward(p1, p2, lengths1, lengths2, idx, 2, grad_dists)
        return grad_p1, grad_p2, None, None, None, None


def ball_query(
    p1: torch.Tensor,
    p2: torch.Tensor,
    lengths1: Union[torch.Tensor, None] = None,
    lengths2: Union[torch.Tensor, None] = None,
    K: int = 500,
    radius: float = 0.2,
    return_nn: bool = True,
):
    """First ``K`` points of ``p2`` (index order) within ``radius`` of each ``p1`` point.

    Same arguments, defaults and return type as the reference
    (functions/ball_query.py:55-142): returns ``KNN(dists, idx, knn)`` with idx
    padded by -1 and dists / knn padded by 0.
    """
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")

    p1 = p1.contiguous()
    p2 = p2.contiguous()
    P1 = p1.shape[1]
    P2 = p2.shape[1]
    N = p1.shape[0]

    if lengths1 is None:
        lengths1 = _full_lengths(N, P1, p1.device)
    if lengths2 is None:
        lengths2 = _full_lengths(N, P2, p1.device)

    dists, idx = _ball_query.apply(p1, p2, lengths1, lengths2, K, radius)
    points_nn = masked_gather(p2, idx) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=points_nn)
