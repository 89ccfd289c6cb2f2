"""ball_query -- the API of the reference's functions/ball_query.py (:55-142) on the HIP kernels.

One autograd node: forward is the `_C.ball_query` operator (csrc/ball_query.hip: index-order scan, or the
cell grid for sparse balls), backward is the KNN backward with the squared-L2 norm over the same neighbour
table (the -1 padding is skipped inside the kernel), as in the reference (:36-52).
"""
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from ._common import neighbor_backward, point_pair
from .knn import _KNN
from .utils import masked_gather


class _BallQueryFn(Function):
    @staticmethod
    def forward(ctx, p1, p2, lengths1, lengths2, K, radius):
        idx, dists = _C.ball_query(p1, p2, lengths1, lengths2, K, radius)
        ctx.mark_non_differentiable(idx)
        ctx.save_for_backward(p1, p2, lengths1, lengths2, idx)
        return dists, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_dists, _grad_idx):
        grad_p1, grad_p2 = neighbor_backward(ctx.saved_tensors, 2, grad_dists, "ball_query backward")
        return grad_p1, grad_p2, None, None, None, None


def ball_query(
    p1: torch.Tensor,
    p2: torch.Tensor,
    lengths1: Optional[torch.Tensor] = None,
    lengths2: Optional[torch.Tensor] = None,
    K: int = 500,
    radius: float = 0.2,
    return_nn: bool = True,
):
    """For every point of `p1` (N, P1, D): the first `K` points of `p2` (N, P2, D), in index order, closer
    than `radius`.

    Returns the reference's named tuple `KNN(dists, idx, knn)`: squared distances (N, P1, K) padded with 0,
    indices (N, P1, K) padded with -1, and -- when `return_nn` -- the gathered neighbours (N, P1, K, D)
    padded with 0 (else None).  `lengths1` / `lengths2` (N,) give the valid points per cloud.
    """
    p1, p2, lengths1, lengths2 = point_pair(p1, p2, lengths1, lengths2)
    if torch.compiler.is_compiling():  # traced graphs see the registered op (pytorch3d_pointops_amd/ops.py)
        idx, dists = torch.ops.pointops_amd.ball_query(p1, p2, lengths1, lengths2, K, radius)
    elif not (torch.is_grad_enabled() and (p1.requires_grad or p2.requires_grad)):
        idx, dists = _C.ball_query(p1, p2, lengths1, lengths2, K, radius)  # nothing to differentiate: no autograd node
    else:
        dists, idx = _BallQueryFn.apply(p1, p2, lengths1, lengths2, K, radius)
    return _KNN(dists=dists, idx=idx, knn=masked_gather(p2, idx) if return_nn else None)
