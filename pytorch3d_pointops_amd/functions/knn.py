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

_KNN = namedtuple("KNN", "dists idx knn")

_full_lengths_cache = {}


def _full_lengths(n: int, p: int, device) -> torch.Tensor:
    """(n,) int64 tensor filled with p: the default `lengths` (reference: functions/knn.py:184-187).
    Read-only for the kernels, so one cached tensor per (n, p, device) saves a fill launch per call."""
    key = (n, p, device)
    t = _full_lengths_cache.get(key)
    if t is None:
        if len(_full_lengths_cache) > 64:
            _full_lengths_cache.clear()
        t = torch.full((n,), p, dtype=torch.int64, device=device)
        _full_lengths_cache[key] = t
    return t


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
        p1, p2, lengths1, lengths2, idx = ctx.saved_tensors
        norm = ctx.norm
        if not (grad_dists.dtype == torch.float32):
            grad_dists = grad_dists.float()
        if not (p1.dtype == torch.float32):
            p1 = p1.float()
        if not (p2.dtype == torch.float32):
            p2 = p2.float()
        grad_p1, grad_p2 = _C.knn_points_backward(p1, p2, lengths1, lengths2, idx, norm, grad_dists)
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
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")

    p1 = p1.contiguous()
    p2 = p2.contiguous()
    P1 = p1.shape[1]
    P2 = p2.shape[1]

    if lengths1 is None:
        lengths1 = _full_lengths(p1.shape[0], P1, p1.device)
    if lengths2 is None:
        lengths2 = _full_lengths(p1.shape[0], P2, p1.device)

    p1_dists, p1_idx = _knn_points.apply(p1, p2, lengths1, lengths2, K, version, norm, return_sorted)

    p2_nn = None
    if return_nn:
        p2_nn = knn_gather(p2, p1_idx, lengths2)

    return _KNN(dists=p1_dists, idx=p1_idx, knn=p2_nn if return_nn else None)


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
        if not (grad_out.dtype == torch.float32):
            grad_out = grad_out.float()
        grad_x = _C.gather_neighbors_backward(grad_out.contiguous(), idx,
                                              lengths if ctx.has_lengths else None, ctx.M)
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
