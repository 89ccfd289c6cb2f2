"""The HIP operators as registered torch ops: ``torch.ops.pointops_amd.*``.

Every callable of the operator boundary (`_C.py`, the mirror of the reference's pybind module
csrc/ext.cpp:15-27) is registered with `torch.library.custom_op` -- schema inferred from the signature, a
fake (meta) implementation that gives output shapes / dtypes without touching the GPU, and autograd formulas
for the differentiable ones (the same closed-form backward kernels the eager wrappers use).  That makes the
ops visible to the dispatcher, `torch.compile` (functions/*.py route through these ops while a graph is
being traced) and the autograd profiler.  The implementations ARE the `_C` ctypes calls into
libpointops_amd.so; there is no second code path.

Eager calls of functions/*.py keep calling `_C` directly inside their autograd nodes: the dispatcher costs
~15 us per call on the host, a quarter of a small (B=2, N=1024) query.
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _C

NS = "pointops_amd"
_op = torch.library.custom_op


def _like(t: Tensor, shape, dtype=None) -> Tensor:
    return t.new_empty(tuple(shape), dtype=dtype or t.dtype)


# --------------------------------------------------------------------------- knn
@_op(f"{NS}::knn_points_idx", mutates_args=())
def knn_points_idx(p1: Tensor, p2: Tensor, lengths1: Tensor, lengths2: Tensor, norm: int, K: int,
                   version: int) -> Tuple[Tensor, Tensor]:
    return _C.knn_points_idx(p1, p2, lengths1, lengths2, norm, K, version)


@knn_points_idx.register_fake
def _(p1, p2, lengths1, lengths2, norm, K, version):
    shape = (p1.shape[0], p1.shape[1], K)
    return _like(p1, shape, torch.int64), _like(p1, shape, torch.float32)


@_op(f"{NS}::knn_points_backward", mutates_args=())
def knn_points_backward(p1: Tensor, p2: Tensor, lengths1: Tensor, lengths2: Tensor, idxs: Tensor, norm: int,
                        grad_dists: Tensor) -> Tuple[Tensor, Tensor]:
    return _C.knn_points_backward(p1, p2, lengths1, lengths2, idxs, norm, grad_dists,
                                  deterministic=torch.are_deterministic_algorithms_enabled())


@knn_points_backward.register_fake
def _(p1, p2, lengths1, lengths2, idxs, norm, grad_dists):
    return _like(p1, p1.shape, torch.float32), _like(p2, p2.shape, torch.float32)


def _neighbor_setup(norm_of):
    def setup(ctx, inputs, output):
        p1, p2, lengths1, lengths2 = inputs[:4]
        ctx.save_for_backward(p1, p2, lengths1, lengths2, output[0])
        ctx.norm = norm_of(inputs)
    return setup


def _neighbor_grad(n_inputs):
    def backward(ctx, _grad_idx, grad_dists):
        p1, p2, lengths1, lengths2, idx = ctx.saved_tensors
        g1, g2 = knn_points_backward(p1.float(), p2.float(), lengths1, lengths2, idx, ctx.norm,
                                     grad_dists.float().contiguous())
        return (g1, g2) + (None,) * (n_inputs - 2)
    return backward


knn_points_idx.register_autograd(_neighbor_grad(7), setup_context=_neighbor_setup(lambda inp: inp[4]))


# --------------------------------------------------------------------------- ball query
@_op(f"{NS}::ball_query", mutates_args=())
def ball_query(p1: Tensor, p2: Tensor, lengths1: Tensor, lengths2: Tensor, K: int,
               radius: float) -> Tuple[Tensor, Tensor]:
    return _C.ball_query(p1, p2, lengths1, lengths2, K, radius)


@ball_query.register_fake
def _(p1, p2, lengths1, lengths2, K, radius):
    shape = (p1.shape[0], p1.shape[1], K)
    return _like(p1, shape, torch.int64), _like(p1, shape, torch.float32)


ball_query.register_autograd(_neighbor_grad(6), setup_context=_neighbor_setup(lambda inp: 2))


# --------------------------------------------------------------------------- farthest point sampling
@_op(f"{NS}::sample_farthest_points", mutates_args=())
def sample_farthest_points(points: Tensor, lengths: Tensor, K: Tensor, start_idxs: Tensor) -> Tensor:
    return _C.sample_farthest_points(points, lengths, K, start_idxs)


@sample_farthest_points.register_fake
def _(points, lengths, K, start_idxs):
    max_k = torch.library.get_ctx().new_dynamic_size()  # max(K): known only on the device
    return _like(points, (points.shape[0], max_k), torch.int64)


# --------------------------------------------------------------------------- packed <-> padded
@_op(f"{NS}::packed_to_padded", mutates_args=())
def packed_to_padded(inputs_packed: Tensor, first_idxs: Tensor, max_size: int) -> Tensor:
    return _C.packed_to_padded(inputs_packed, first_idxs, max_size)


@packed_to_padded.register_fake
def _(inputs_packed, first_idxs, max_size):
    return _like(inputs_packed, (first_idxs.shape[0], max_size, inputs_packed.shape[1]))


@_op(f"{NS}::padded_to_packed", mutates_args=())
def padded_to_packed(inputs_padded: Tensor, first_idxs: Tensor, num_inputs: int) -> Tensor:
    return _C.padded_to_packed(inputs_padded, first_idxs, num_inputs)


@padded_to_packed.register_fake
def _(inputs_padded, first_idxs, num_inputs):
    return _like(inputs_padded, (num_inputs, inputs_padded.shape[2]))


def _p2p_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1])
    ctx.rows = inputs[0].shape[0]


def _pad_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1])
    ctx.max_size = inputs[0].shape[1]


packed_to_padded.register_autograd(
    lambda ctx, g: (padded_to_packed(g.contiguous(), ctx.saved_tensors[0], ctx.rows), None, None),
    setup_context=_p2p_setup)
padded_to_packed.register_autograd(
    lambda ctx, g: (packed_to_padded(g.contiguous(), ctx.saved_tensors[0], ctx.max_size), None, None),
    setup_context=_pad_setup)


# --------------------------------------------------------------------------- neighbour gather
@_op(f"{NS}::gather_neighbors", mutates_args=())
def gather_neighbors(x: Tensor, idx: Tensor, lengths: Optional[Tensor]) -> Tensor:
    return _C.gather_neighbors(x, idx, lengths)


@gather_neighbors.register_fake
def _(x, idx, lengths):
    return _like(x, (idx.shape[0], idx.shape[1], idx.shape[2], x.shape[2]), torch.float32)


@_op(f"{NS}::gather_neighbors_backward", mutates_args=())
def gather_neighbors_backward(grad_out: Tensor, idx: Tensor, lengths: Optional[Tensor], M: int) -> Tensor:
    return _C.gather_neighbors_backward(grad_out, idx, lengths, M,
                                        deterministic=torch.are_deterministic_algorithms_enabled())


@gather_neighbors_backward.register_fake
def _(grad_out, idx, lengths, M):
    return _like(grad_out, (grad_out.shape[0], M, grad_out.shape[3]), torch.float32)


def _gather_setup(ctx, inputs, output):
    x, idx, lengths = inputs
    ctx.has_lengths = lengths is not None
    ctx.save_for_backward(idx, *([lengths] if lengths is not None else []))
    ctx.M = x.shape[1]


def _gather_grad(ctx, grad_out):
    idx = ctx.saved_tensors[0]
    lengths = ctx.saved_tensors[1] if ctx.has_lengths else None
    return gather_neighbors_backward(grad_out.float().contiguous(), idx, lengths, ctx.M), None, None


gather_neighbors.register_autograd(_gather_grad, setup_context=_gather_setup)


# --------------------------------------------------------------------------- covariances
@_op(f"{NS}::point_covariances", mutates_args=())
def point_covariances(knn: Tensor) -> Tensor:
    return _C.point_covariances(knn)


@point_covariances.register_fake
def _(knn):
    return _like(knn, (knn.shape[0], knn.shape[1], knn.shape[3], knn.shape[3]))


@_op(f"{NS}::point_covariances_backward", mutates_args=())
def point_covariances_backward(knn: Tensor, grad_cov: Tensor) -> Tensor:
    return _C.point_covariances_backward(knn, grad_cov)


@point_covariances_backward.register_fake
def _(knn, grad_cov):
    return _like(knn, knn.shape)


point_covariances.register_autograd(
    lambda ctx, g: point_covariances_backward(ctx.saved_tensors[0], g.float().contiguous()),
    setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0].contiguous()))


# --------------------------------------------------------------------------- chamfer
@_op(f"{NS}::chamfer_reduce", mutates_args=())
def chamfer_reduce(dists: Tensor, lengths: Tensor, weights: Optional[Tensor], mean: bool) -> Tensor:
    return _C.chamfer_reduce(dists, lengths, weights, mean)


@chamfer_reduce.register_fake
def _(dists, lengths, weights, mean):
    return _like(dists, (dists.shape[0],), torch.float32)


@_op(f"{NS}::chamfer_forward", mutates_args=())
def chamfer_forward(dists: Tensor, idx: Tensor, x_lengths: Tensor, y_lengths: Tensor, weights: Optional[Tensor],
                    x_feats: List[Tensor], y_feats: List[Tensor], abs_cosine: bool, mean: bool) -> Tensor:
    return _C.chamfer_forward(dists, idx, x_lengths, y_lengths, weights, list(x_feats), list(y_feats), abs_cosine,
                              mean)


@chamfer_forward.register_fake
def _(dists, idx, x_lengths, y_lengths, weights, x_feats, y_feats, abs_cosine, mean):
    return _like(dists, (1 + len(x_feats), dists.shape[0]), torch.float32)


@_op(f"{NS}::chamfer_backward", mutates_args=())
def chamfer_backward(x: Tensor, y: Tensor, idx: Tensor, x_lengths: Tensor, y_lengths: Tensor,
                     weights: Optional[Tensor], grad_out: Tensor, norm: int, x_feats: List[Tensor],
                     y_feats: List[Tensor], abs_cosine: bool, mean: bool) -> List[Tensor]:
    gx, gy, gxf, gyf = _C.chamfer_backward(x, y, idx, x_lengths, y_lengths, weights, grad_out, norm, list(x_feats),
                                           list(y_feats), abs_cosine, mean)
    return [gx, gy, *gxf, *gyf]  # grad_x, grad_y, then the x feature grads, then the y feature grads


@chamfer_backward.register_fake
def _(x, y, idx, x_lengths, y_lengths, weights, grad_out, norm, x_feats, y_feats, abs_cosine, mean):
    return [torch.empty_like(t) for t in (x, y, *x_feats, *y_feats)]


# --------------------------------------------------------------------------- sample_pdf (in place)
@_op(f"{NS}::sample_pdf", mutates_args=("outputs",))
def sample_pdf(bins: Tensor, weights: Tensor, outputs: Tensor, eps: float) -> None:
    _C.sample_pdf(bins, weights, outputs, eps)


@sample_pdf.register_fake
def _(bins, weights, outputs, eps):
    return None


def registered_ops():
    """Names of the registered operators (tests check them against `_C`)."""
    return sorted(n for n in dir(getattr(torch.ops, NS)) if not n.startswith("_") and n != "name")
