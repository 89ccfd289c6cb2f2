"""knn_points / knn_gather -- same API as the reference's functions/knn.py.

reference: pytorch3d_pointops/functions/knn.py:18 (_KNN), :21-111 (_knn_points),
:114-197 (knn_points), :200-250 (knn_gather).
"""
from collections import namedtuple
from typing import Union

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from ._common import deterministic_requested, as_f32, full_lengths, neighbor_backward, point_pair

_KNN = namedtuple("KNN", "dists idx knn")


class _knn_points(Function):
    """autograd wrapper around the HIP KNN kernels (reference: functions/knn.py:21-111).

    The HIP kernel already emits each row in ascending (dist, idx) order -- the
    order of the reference CPU kernel (csrc/knn/knn_cpu.cpp:59-65) -- so the
    reference's device-side ``sort`` + ``gather`` pass (functions/knn.py:77-89) and
    its ``lengths2.min()`` host sync are not needed; ``return_sorted`` is accepted
    and, as on the reference CPU path, changes nothing.
    """

    @staticmethod
    def forward(ctx, p1, p2, lengths1, lengths2, K, version, norm: int = 2,
                return_sorted: bool = True):
        if not ((norm == 1) or (norm == 2)):
            raise ValueError("Support for 1 or 2 norm.")
        idx, dists = _C.knn_points_idx(p1, p2, lengths1, lengths2, norm, K, version)
        ctx.save_for_backward(p1, p2, lengths1, lengths2, idx)
        ctx.mark_non_differentiable(idx)
        ctx.norm = norm
        return dists, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_dists, grad_idx):
        # grad_p2 is a scatter-add with fp32 atomics: the reference's CUDA backward raises the same
        # nondeterminism alert (csrc/knn/knn.cu:538)
        grad_p1, grad_p2 = neighbor_backward(ctx.saved_tensors, ctx.norm, grad_dists, "knn_points backward")
        return grad_p1, grad_p2, None, None, None, None, None, None


def knn_points(
    p1: torch.Tensor,
    p2: torch.Tensor,
    lengths1: Union[torch.Tensor, None] = None,
    lengths2: Union[torch.Tensor, None] = None,
    norm: int = 2,
    K: int = 1,
    version: int = -1,
    return_nn: bool = False,
    return_sorted: bool = True,
) -> _KNN:
    """K nearest neighbours of every point of ``p1`` in ``p2`` (per cloud).

    Same arguments, defaults, return type and padding as the reference
    (functions/knn.py:114-197): ``p1`` (N,P1,D), ``p2`` (N,P2,D), optional int64
    ``lengths1``/``lengths2`` (N,); returns ``KNN(dists (N,P1,K), idx (N,P1,K),
    knn (N,P1,K,D) or None)``; dists/idx are zero for rows >= lengths1[n] and
    slots >= lengths2[n].
    """
    p1, p2, lengths1, lengths2 = point_pair(p1, p2, lengths1, lengths2)
    if torch.compiler.is_compiling():  # a graph is being traced: the registered op (pytorch3d_pointops_amd/ops.py)
        if norm not in (1, 2):
            raise ValueError("Support for 1 or 2 norm.")
        idx, dists = torch.ops.pointops_amd.knn_points_idx(p1, p2, lengths1, lengths2, norm, K, version)
    elif not (torch.is_grad_enabled() and (p1.requires_grad or p2.requires_grad)):
        # nothing to differentiate: no autograd node (6 us of a 57 us call at B=2, N=1024)
        if not ((norm == 1) or (norm == 2)):
            raise ValueError("Support for 1 or 2 norm.")
        idx, dists = _C.knn_points_idx(p1, p2, lengths1, lengths2, norm, K, version)
    else:
        dists, idx = _knn_points.apply(p1, p2, lengths1, lengths2, K, version, norm, return_sorted)
    return _KNN(dists=dists, idx=idx, knn=knn_gather(p2, idx, lengths2) if return_nn else None)


class _gather_neighbors(Function):
    """out[n,l,k,:] = x[n, idx[n,l,k], :] with knn (k >= lengths) and -1 masks."""

    @staticmethod
    def forward(ctx, x, idx, lengths):
        out = _C.gather_neighbors(x, idx, lengths)
        ctx.save_for_backward(idx, lengths if lengths is not None else idx.new_empty(0))
        ctx.has_lengths = lengths is not None
        ctx.M = x.shape[1]
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        idx, lengths = ctx.saved_tensors
        # fp32 scatter-add (LDS or device atomics); the inverted-table form when determinism is requested
        grad_x = _C.gather_neighbors_backward(as_f32(grad_out).contiguous(), idx,
                                              lengths if ctx.has_lengths else None, ctx.M,
                                              deterministic=deterministic_requested())
        return grad_x, None, None


def knn_gather(x: torch.Tensor, idx: torch.Tensor, lengths: Union[torch.Tensor, None] = None):
    """Index ``x`` (N,M,U) with ``idx`` (N,L,K) from knn_points -> (N,L,K,U).

    Same contract as the reference (functions/knn.py:200-250):
    ``x_out[n,l,k] = x[n, idx[n,l,k]]``, zero where ``k >= lengths[n]``;
    differentiable w.r.t. ``x``.  One fused HIP gather (and one scatter-add for the
    backward) replaces expand + torch.gather + masked fill and its host sync.
    """
    N, M, U = x.shape
    _N, L, K = idx.shape
    if N != _N:
        raise ValueError("x and idx must have same batch dimension.")
    if x.dtype != torch.float32:
        # rare path (the reference gathers any dtype): same semantics through torch
        return _knn_gather_torch(x, idx, lengths)
    if torch.compiler.is_compiling():
        return torch.ops.pointops_amd.gather_neighbors(x, idx, lengths)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return _C.gather_neighbors(x, idx, lengths)  # nothing to differentiate: no autograd node
    return _gather_neighbors.apply(x, idx, lengths)


def _knn_gather_torch(x, idx, lengths):
    N, M, U = x.shape
    _N, L, K = idx.shape
    if lengths is None:
        lengths = torch.full((N,), M, dtype=torch.int64, device=x.device)
    x_out = x[:, :, None].expand(-1, -1, K, -1).gather(1, idx[:, :, :, None].expand(-1, -1, -1, U))
    mask = lengths[:, None] <= torch.arange(K, device=x.device)[None]
    mask = mask[:, None, :, None].expand(-1, L, -1, U)
    return x_out.masked_fill(mask, 0.0)
