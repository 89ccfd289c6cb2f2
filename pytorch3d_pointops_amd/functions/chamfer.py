"""chamfer_distance -- same API as the reference's functions/chamfer.py.

reference: pytorch3d_pointops/functions/chamfer.py:17-35 (reduction validation),
:38-82 (_handle_pointcloud_input), :85-189 (_chamfer_distance_single_direction),
:192-214 (_apply_batch_reduction), :217-365 (chamfer_distance).

Structure on MI355X: each direction is one K=1 HIP KNN scan (register top-1,
scalar-path streaming of the other cloud) followed by ONE fused masked
reduction kernel (mask rows >= lengths, sum over points, * weights,
/ clamp(lengths,1)) with a closed-form custom backward, instead of the reference's
chain of ~6 elementwise/reduction torch kernels and host syncs per direction.  The
cosine feature term gathers neighbour features with the fused HIP gather.
"""
from typing import Union

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from .knn import knn_gather, knn_points


def _is_pointclouds(obj) -> bool:
    # The reference tests isinstance(points, Pointclouds) (functions/chamfer.py:49).
    # Accept our own container and any Pointclouds-like object exposing the three
    # accessors the reference calls (points_padded / num_points_per_cloud / features_padded).
    return (
        not torch.is_tensor(obj)
        and hasattr(obj, "points_padded")
        and hasattr(obj, "num_points_per_cloud")
        and hasattr(obj, "features_padded")
    )


def _validate_chamfer_reduction_inputs(
    batch_reduction: Union[str, None], point_reduction: Union[str, None]
) -> None:
    if batch_reduction is not None and batch_reduction not in ["mean", "sum"]:
        raise ValueError('batch_reduction must be one of ["mean", "sum"] or None')
    if point_reduction is not None and point_reduction not in ["mean", "sum", "max"]:
        raise ValueError('point_reduction must be one of ["mean", "sum", "max"] or None')
    if point_reduction is None and batch_reduction is not None:
        raise ValueError("Batch reduction must be None if point_reduction is None")


def _handle_pointcloud_input(points, lengths, features):
    if _is_pointclouds(points):
        X = points.points_padded()
        lengths = points.num_points_per_cloud()
        features = points.features_padded()  # dict (possibly empty)
    elif torch.is_tensor(points):
        if points.ndim != 3:
            raise ValueError("Expected points to be of shape (N, P, D)")
        X = points
        if lengths is not None:
            if lengths.ndim != 1 or lengths.shape[0] != X.shape[0]:
                raise ValueError("Expected lengths to be of shape (N,)")
            if lengths.max() > X.shape[1]:
                raise ValueError("A length value was too long")
        if lengths is None:
            lengths = torch.full((X.shape[0],), X.shape[1], dtype=torch.int64, device=points.device)
        if features is not None:
            if isinstance(features, dict):
                for feature_name, feature_tensor in features.items():
                    if feature_tensor is not None and feature_tensor.ndim != 3:
                        raise ValueError(f"Expected {feature_name} to be of shape (N, P, C)")
            elif torch.is_tensor(features) and features.ndim != 3:
                raise ValueError("Expected features to be of shape (N, P, C)")
    else:
        raise ValueError(
            "The input pointclouds should be either "
            + "Pointclouds objects or torch.Tensor of shape "
            + "(minibatch, num_points, 3)."
        )
    return X, lengths, features


class _masked_point_reduce(Function):
    """(N,P) per-point terms -> (N,) masked sum [* weights] [/ clamp(lengths,1)].

    Forward is the fused HIP reduction; backward is its closed form
    ``grad_in[n,i] = g[n] * w[n] / clamp(len[n],1)`` for ``i < len[n]`` else 0.
    """

    @staticmethod
    def forward(ctx, terms, lengths, weights, mean: bool):
        out = _C.chamfer_reduce(terms, lengths, weights, mean)
        ctx.save_for_backward(lengths, weights if weights is not None else terms.new_empty(0))
        ctx.has_w = weights is not None
        ctx.mean = mean
        ctx.P = terms.shape[1]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lengths, weights = ctx.saved_tensors
        scale = g.float()
        if ctx.has_w:
            scale = scale * weights
        if ctx.mean:
            scale = scale / lengths.clamp(min=1)
        mask = torch.arange(ctx.P, device=g.device)[None] < lengths[:, None]
        return scale[:, None] * mask, None, None, None


class _chamfer_direction(Function):
    """One chamfer direction for point_reduction in {"sum","mean"}: K=1 grid search + ONE fused
    kernel for the masked / weighted / normalised per-cloud sums of the point term and of every
    cosine feature term, with a closed-form fused backward (device half: csrc/chamfer.hip).
    Returns a (1+F, N) tensor: row 0 the point term, row 1+f feature f."""

    @staticmethod
    def forward(ctx, x, y, x_lengths, y_lengths, weights, norm, mean, abs_cosine, *feats):
        F_ = len(feats) // 2
        x_feats = [f.contiguous() for f in feats[:F_]]
        y_feats = [f.contiguous() for f in feats[F_:]]
        x, y = x.contiguous(), y.contiguous()
        idx, dists = _C.knn_points_idx(x, y, x_lengths, y_lengths, norm, 1, -1)
        out = _C.chamfer_forward(dists.view(dists.shape[0], dists.shape[1]), idx.view(idx.shape[0], idx.shape[1]),
                                 x_lengths, y_lengths, weights, x_feats, y_feats, abs_cosine, mean)
        ctx.save_for_backward(x, y, idx, x_lengths, y_lengths,
                              weights if weights is not None else x.new_empty(0), *x_feats, *y_feats)
        ctx.has_w = weights is not None
        ctx.cfg = (int(norm), bool(mean), bool(abs_cosine), F_)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        norm, mean, abs_cosine, F_ = ctx.cfg
        x, y, idx, xl, yl, w = ctx.saved_tensors[:6]
        x_feats = list(ctx.saved_tensors[6:6 + F_])
        y_feats = list(ctx.saved_tensors[6 + F_:6 + 2 * F_])
        gx, gy, gxf, gyf = _C.chamfer_backward(x, y, idx.view(idx.shape[0], idx.shape[1]), xl, yl,
                                               w if ctx.has_w else None, grad_out.contiguous().float(), norm,
                                               x_feats, y_feats, abs_cosine, mean)
        return (gx, gy, None, None, None, None, None, None, *gxf, *gyf)


def _fused_direction_ok(x, y, x_features, y_features, feature_names, return_features, point_reduction):
    if point_reduction not in ("sum", "mean"):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32):
        return False
    if return_features:
        if len(feature_names) > _C.CHAMFER_MAX_FEATURES:
            return False
        for name in feature_names:
            a, b = x_features[name], y_features[name]
            if a.dtype != torch.float32 or b.dtype != torch.float32 or a.ndim != 3 or b.ndim != 3:
                return False
            if a.shape[2] != b.shape[2] or a.shape[2] > _C.CHAMFER_MAX_CHANNELS:
                return False
            if a.shape[:2] != x.shape[:2] or b.shape[:2] != y.shape[:2]:
                return False
    return True


def _chamfer_distance_single_direction(
    x,
    y,
    x_lengths,
    y_lengths,
    x_features,
    y_features,
    weights,
    point_reduction: Union[str, None],
    norm: int,
    abs_cosine: bool,
    feature_names: Union[list, None] = None,
):
    if feature_names and x_features is not None and y_features is not None:
        for feature_name in feature_names:
            if feature_name not in x_features:
                raise ValueError(f"Feature '{feature_name}' is missing in x_features.")
            if feature_name not in y_features:
                raise ValueError(f"Feature '{feature_name}' is missing in y_features.")

    return_features = (
        x_features is not None
        and y_features is not None
        and feature_names is not None
        and len(feature_names) > 0
    )

    N, P1, D = x.shape
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")
    if weights is not None:
        if weights.size(0) != N:
            raise ValueError("weights must be of shape (N,).")
        if not (weights >= 0).all():
            raise ValueError("weights cannot be negative.")
        if weights.sum() == 0.0:
            # All-zero weights: a zero loss that stays attached to x's graph.  The reference's
            # early return (functions/chamfer.py:128-130) hands back a TENSOR where its caller
            # expects the per-feature dict and raises IndexError for every reduction mode
            # (verified against the compiled reference); upstream PyTorch3D's intent -- zeros --
            # is what is implemented here ("Deviations").
            zero = (x.sum((1, 2)) * weights) * 0.0  # (N,)
            if point_reduction is None:
                zero = zero[:, None].expand(N, P1)
            zf = {name: zero for name in feature_names} if return_features else None
            return zero, zf

    if _fused_direction_ok(x, y, x_features, y_features, feature_names, return_features, point_reduction):
        names = list(feature_names) if return_features else []
        feats = [x_features[k] for k in names] + [y_features[k] for k in names]
        out = _chamfer_direction.apply(x, y, x_lengths, y_lengths,
                                       weights.to(torch.float32) if weights is not None else None, norm,
                                       point_reduction == "mean", abs_cosine, *feats)
        return out[0], ({k: out[1 + i] for i, k in enumerate(names)} if return_features else None)

    x_nn = knn_points(x, y, lengths1=x_lengths, lengths2=y_lengths, norm=norm, K=1)
    cham_x = x_nn.dists[..., 0]  # (N, P1); rows >= x_lengths are already 0 (kernel padding)

    fused = point_reduction in ("sum", "mean") and cham_x.dtype == torch.float32
    w32 = None
    if weights is not None:
        w32 = weights.to(torch.float32)

    cham_features_x = None
    if not fused:
        x_mask = torch.arange(P1, device=x.device)[None] >= x_lengths[:, None]  # (N, P1)
        cham_x = cham_x.masked_fill(x_mask, 0.0)
        if weights is not None:
            cham_x = cham_x * weights.view(N, 1)

    if return_features:
        cham_features_x = {}
        for feature_name in feature_names:
            x_feature = x_features[feature_name]
            y_feature = y_features[feature_name]
            x_feature_near = knn_gather(y_feature, x_nn.idx, y_lengths)[..., 0, :]
            cosine_sim = F.cosine_similarity(x_feature, x_feature_near, dim=2, eps=1e-6)
            cosine_sim = torch.abs(cosine_sim) if abs_cosine else cosine_sim
            feature_distance = 1 - cosine_sim  # (N, P1)
            if fused:
                cham_features_x[feature_name] = _masked_point_reduce.apply(
                    feature_distance.contiguous(), x_lengths, w32, point_reduction == "mean")
            else:
                feature_distance = feature_distance.masked_fill(x_mask, 0.0)
                if weights is not None:
                    feature_distance = feature_distance * weights.view(N, 1)
                cham_features_x[feature_name] = feature_distance

    if point_reduction == "max":
        assert not return_features
        cham_x = cham_x.max(1).values  # (N,)
    elif fused:
        cham_x = _masked_point_reduce.apply(cham_x.contiguous(), x_lengths, w32,
                                            point_reduction == "mean")
    return cham_x, cham_features_x


def _apply_batch_reduction(cham_x, cham_features_x, weights, batch_reduction: Union[str, None]):
    if batch_reduction is None:
        return (cham_x, cham_features_x)
    N = cham_x.shape[0]
    cham_x = cham_x.sum()
    if cham_features_x is not None:
        for feature_name in cham_features_x:
            cham_features_x[feature_name] = cham_features_x[feature_name].sum()
    if batch_reduction == "mean":
        if weights is None:
            div = max(N, 1)
        elif weights.sum() == 0.0:
            div = 1
        else:
            div = weights.sum()
        cham_x = cham_x / div
        if cham_features_x is not None:
            for feature_name in cham_features_x:
                cham_features_x[feature_name] = cham_features_x[feature_name] / div
    return (cham_x, cham_features_x)


def chamfer_distance(
    x,
    y,
    x_lengths=None,
    y_lengths=None,
    x_features=None,
    y_features=None,
    weights=None,
    batch_reduction: Union[str, None] = "mean",
    point_reduction: Union[str, None] = "mean",
    norm: int = 2,
    single_directional: bool = False,
    abs_cosine: bool = True,
    feature_names: Union[list, None] = None,
):
    """Chamfer distance between two batches of clouds; same arguments, defaults,
    error behaviour and return structure ``(loss, loss_features)`` as the reference
    (functions/chamfer.py:217-365).  ``x`` / ``y`` are (N,P,D) tensors or
    Pointclouds-like objects; ``loss_features`` is None or a dict per feature name.
    """
    _validate_chamfer_reduction_inputs(batch_reduction, point_reduction)

    if not ((norm == 1) or (norm == 2)):
        raise ValueError("Support for 1 or 2 norm.")

    if point_reduction == "max" and (feature_names is not None and len(feature_names) > 0):
        raise ValueError('Features must be None if point_reduction is "max"')

    x, x_lengths, x_features = _handle_pointcloud_input(x, x_lengths, x_features)
    y, y_lengths, y_features = _handle_pointcloud_input(y, y_lengths, y_features)

    cham_x, cham_features_x = _chamfer_distance_single_direction(
        x, y, x_lengths, y_lengths, x_features, y_features, weights, point_reduction, norm,
        abs_cosine, feature_names,
    )
    if single_directional:
        loss = cham_x
        loss_features = cham_features_x
    else:
        cham_y, cham_features_y = _chamfer_distance_single_direction(
            y, x, y_lengths, x_lengths, y_features, x_features, weights, point_reduction, norm,
            abs_cosine, feature_names,
        )
        if point_reduction == "max":
            loss = torch.maximum(cham_x, cham_y)
            loss_features = None
        elif point_reduction is not None:
            loss = cham_x + cham_y
            if cham_features_x is not None:
                loss_features = {}
                for feature_name in cham_features_x:
                    if feature_name in cham_features_y:
                        loss_features[feature_name] = (
                            cham_features_x[feature_name] + cham_features_y[feature_name]
                        )
                    else:
                        loss_features[feature_name] = cham_features_x[feature_name]
            else:
                loss_features = None
        else:
            loss = (cham_x, cham_y)
            if cham_features_x is not None:
                loss_features = {}
                for feature_name in cham_features_x:
                    if feature_name in cham_features_y:
                        loss_features[feature_name] = (
                            cham_features_x[feature_name],
                            cham_features_y[feature_name],
                        )
                    else:
                        loss_features[feature_name] = (cham_features_x[feature_name], None)
            else:
                loss_features = None
    return _apply_batch_reduction(loss, loss_features, weights, batch_reduction)
