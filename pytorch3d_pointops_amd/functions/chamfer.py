"""chamfer_distance -- the API of the reference's functions/chamfer.py (:217-365) on the HIP kernels.

Structure on MI355X.  A direction (x -> y) is ONE autograd node `_chamfer_direction`: the K=1 exact grid
search, then one fused kernel for the masked / weighted / length-normalised per-cloud sums of the point term
and of every cosine feature term, with a closed-form fused backward (csrc/chamfer.hip) -- instead of the
reference's chain of ~25 torch kernels and five host syncs per direction (:85-189).  Reductions the fused
kernel does not cover (point_reduction "max" / None, exotic feature shapes) take the composed path
`_direction_composed`, built from knn_points / knn_gather and the fused masked reduction `_masked_point_reduce`.

Layout of this module: `_Terms` carries what one direction produces (the point term and a dict of feature
terms); `_combine` merges the two directions per reduction mode; `_reduce_batch` applies the batch reduction
to every term alike.  Error texts are the reference's (:29-35, :57-80, :120-126, :291-296).
"""
from typing import Dict, NamedTuple, Optional, Union

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from ._common import alert_not_deterministic, full_lengths, lengths_max
from .knn import knn_gather, knn_points


class _Terms(NamedTuple):
    """Loss terms of one direction (or of both, combined): `points` and one entry per feature name
    (None = no feature loss requested).  Shapes: (N,) after a point reduction, (N, P) without one."""
    points: object
    feats: Optional[Dict[str, object]]

    def map(self, fn):
        return _Terms(fn(self.points), None if self.feats is None else {k: fn(v) for k, v in self.feats.items()})


def _is_pointclouds(obj) -> bool:
    # The reference tests isinstance(points, Pointclouds) (functions/chamfer.py:49).  Our own container and
    # any object with the three accessors the reference calls qualify.
    return not torch.is_tensor(obj) and all(
        hasattr(obj, a) for a in ("points_padded", "num_points_per_cloud", "features_padded"))


def _validate_chamfer_reduction_inputs(batch_reduction: Union[str, None], point_reduction: Union[str, None]) -> None:
    allowed = {"batch_reduction": (batch_reduction, ("mean", "sum")),
               "point_reduction": (point_reduction, ("mean", "sum", "max"))}
    for name, (value, modes) in allowed.items():
        if value is not None and value not in modes:
            raise ValueError(f"{name} must be one of [{', '.join(chr(34) + m + chr(34) for m in modes)}] or None")
    if point_reduction is None and batch_reduction is not None:
        raise ValueError("Batch reduction must be None if point_reduction is None")


def _feature_dims_ok(features) -> None:
    if isinstance(features, dict):
        for name, t in features.items():
            if t is not None and t.ndim != 3:
                raise ValueError(f"Expected {name} to be of shape (N, P, C)")
    elif torch.is_tensor(features) and features.ndim != 3:
        raise ValueError("Expected features to be of shape (N, P, C)")


def _handle_pointcloud_input(points, lengths, features):
    """(padded points (N,P,D), lengths (N,), features) of a tensor or Pointclouds-like argument."""
    if _is_pointclouds(points):
        return points.points_padded(), points.num_points_per_cloud(), points.features_padded()
    if not torch.is_tensor(points):
        raise ValueError("The input pointclouds should be either Pointclouds objects or torch.Tensor of shape "
                         "(minibatch, num_points, 3).")
    if points.ndim != 3:
        raise ValueError("Expected points to be of shape (N, P, D)")
    n, p = points.shape[:2]
    if lengths is None:
        # (one read-only tensor per (n, p, device): no fill launch per call, and the SAME tensor call after call --
        # what the grid reuse of `_C.knn_points_idx` recognises a target by)
        lengths = torch.full((n,), p, dtype=torch.int64, device=points.device) if torch.compiler.is_compiling() \
            else full_lengths(n, p, points.device)
    else:
        if lengths.ndim != 1 or lengths.shape[0] != n:
            raise ValueError("Expected lengths to be of shape (N,)")
        if (lengths.max() if torch.compiler.is_compiling() else lengths_max(lengths)) > p:
            raise ValueError("A length value was too long")
    if features is not None:
        _feature_dims_ok(features)
    return points, lengths, features


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
        alert_not_deterministic("chamfer_distance backward")  # grad_y / grad_y_feats: fp32 atomics
        gx, gy, gxf, gyf = _C.chamfer_backward(x, y, idx.view(idx.shape[0], idx.shape[1]), xl, yl,
                                               w if ctx.has_w else None, grad_out.contiguous().float(), norm,
                                               x_feats, y_feats, abs_cosine, mean)
        return (gx, gy, None, None, None, None, None, None, *gxf, *gyf)


class _chamfer_pair(Function):
    """BOTH directions, their sum and the batch reduction as one autograd node (no weights, point_reduction in
    {"sum","mean"}): the composed form spends ~20 launches on (1+F, N)-sized tensors per call -- selects, adds, sums,
    divisions and their backward twins -- which cost as much as a K=1 search at the cfg4 size.  Returns 1+F tensors
    (loss, then one per feature), each () after a batch reduction or (N,) without one."""

    @staticmethod
    def forward(ctx, x, y, x_lengths, y_lengths, norm, mean, abs_cosine, batch_reduction, *feats):
        F_ = len(feats) // 2
        x_feats = [f.contiguous() for f in feats[:F_]]
        y_feats = [f.contiguous() for f in feats[F_:]]
        x, y = x.contiguous(), y.contiguous()
        N = x.shape[0]
        ctx.cfg = (int(norm), bool(mean), bool(abs_cosine), batch_reduction, F_)
        ctx.native = not _C.grid_cache_enabled()
        if ctx.native:  # one native call (the opt-in grid cache keeps the composed calls: its bookkeeping is Python's)
            outs, idx_xy, idx_yx = _C.chamfer_pair_forward(x, y, x_lengths, y_lengths, norm, x_feats, y_feats,
                                                           abs_cosine, mean, batch_reduction)
            ctx.save_for_backward(x, y, idx_xy, idx_yx, x_lengths, y_lengths, *x_feats, *y_feats)
            return tuple(outs)

        def direction(a, b, a_len, b_len, a_feats, b_feats):
            idx, dists = _C.knn_points_idx(a, b, a_len, b_len, norm, 1, -1)
            idx = idx.view(idx.shape[0], idx.shape[1])
            return idx, _C.chamfer_forward(dists.view(idx.shape), idx, a_len, b_len, None, a_feats, b_feats,
                                           abs_cosine, mean)

        idx_xy, rows = direction(x, y, x_lengths, y_lengths, x_feats, y_feats)
        idx_yx, rows_yx = direction(y, x, y_lengths, x_lengths, y_feats, x_feats)
        rows += rows_yx  # (1+F, N)
        if batch_reduction is not None:
            rows = rows.sum(1)
            if batch_reduction == "mean":
                rows /= max(N, 1)
        ctx.save_for_backward(x, y, idx_xy, idx_yx, x_lengths, y_lengths, *x_feats, *y_feats)
        return tuple(r.clone() for r in rows.unbind(0))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        norm, mean, abs_cosine, batch_reduction, F_ = ctx.cfg
        x, y, idx_xy, idx_yx, xl, yl = ctx.saved_tensors[:6]
        x_feats = list(ctx.saved_tensors[6:6 + F_])
        y_feats = list(ctx.saved_tensors[6 + F_:6 + 2 * F_])
        N = x.shape[0]
        alert_not_deterministic("chamfer_distance backward")  # grad of the TARGET cloud: fp32 atomics
        if ctx.native:
            gx, gy, gxf, gyf = _C.chamfer_pair_backward(x, y, idx_xy, idx_yx, xl, yl, grads, norm, x_feats, y_feats,
                                                        abs_cosine, mean, batch_reduction)
            return (gx, gy, None, None, None, None, None, None, *gxf, *gyf)
        like = next(g for g in grads if g is not None)
        g = torch.stack([torch.zeros_like(like) if gi is None else gi for gi in grads]).float()
        if batch_reduction is not None:  # (1+F,) -> the same value for every cloud
            if batch_reduction == "mean":
                g = g / max(N, 1)
            g = g[:, None].expand(1 + F_, N)
        g = g.contiguous()
        gx, gy, gxf, gyf = _C.chamfer_backward(x, y, idx_xy, xl, yl, None, g, norm, x_feats, y_feats, abs_cosine,
                                               mean)
        # the reverse direction ADDS into the same buffers (its dense terms into gy, its atomics into gx): no second
        # set of gradients, no zero fills, no sums of (N, P, D) tensors
        _C.chamfer_backward(y, x, idx_yx, yl, xl, None, g, norm, y_feats, x_feats, abs_cosine, mean,
                            into=(gy, gx, gyf, gxf))
        return (gx, gy, None, None, None, None, None, None, *gxf, *gyf)


def _fused_direction_ok(x, y, x_features, y_features, feature_names, return_features, point_reduction):
    if point_reduction not in ("sum", "mean") or torch.compiler.is_compiling():
        return False  # (traced graphs take the composed path over the registered ops)
    if torch.are_deterministic_algorithms_enabled():
        return False  # the fused backward scatters with fp32 atomics; the composed path has deterministic backward passes
    if not (x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32):
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


def _direction_composed(x, y, x_lengths, y_lengths, x_features, y_features, names, weights, point_reduction, norm,
                        abs_cosine) -> _Terms:
    """A direction out of knn_points / knn_gather / cosine_similarity and the fused masked reduction: every
    reduction mode, any feature width."""
    N, P1 = x.shape[:2]
    nn = knn_points(x, y, lengths1=x_lengths, lengths2=y_lengths, norm=norm, K=1)
    per_point = {"": nn.dists[..., 0]}  # (N, P1); rows >= x_lengths are already 0 (kernel padding)
    for name in names:
        nearest = knn_gather(y_features[name], nn.idx, y_lengths)[..., 0, :]
        cos = F.cosine_similarity(x_features[name], nearest, dim=2, eps=1e-6)
        per_point[name] = 1 - (cos.abs() if abs_cosine else cos)

    if point_reduction in ("sum", "mean") and per_point[""].dtype == torch.float32 and not torch.compiler.is_compiling():
        w32 = None if weights is None else weights.to(torch.float32)
        done = {k: _masked_point_reduce.apply(v.contiguous(), x_lengths, w32, point_reduction == "mean")
                for k, v in per_point.items()}
    else:
        outside = torch.arange(P1, device=x.device)[None] >= x_lengths[:, None]  # (N, P1) padding mask
        done = {}
        for k, v in per_point.items():
            v = v.masked_fill(outside, 0.0)
            done[k] = v if weights is None else v * weights.view(N, 1)
        if point_reduction == "max":
            assert not names
            done[""] = done[""].max(1).values
    return _Terms(done[""], {k: done[k] for k in names} if names else None)


def _direction_names(x, y, x_features, y_features, feature_names) -> list:
    """The feature names a direction x -> y evaluates, after the reference's argument checks
    (functions/chamfer.py:100-118): every requested name present on both sides, batch and point dimension equal."""
    have_both = x_features is not None and y_features is not None
    if feature_names and have_both:
        for name in feature_names:
            for side, table in (("x_features", x_features), ("y_features", y_features)):
                if name not in table:
                    raise ValueError(f"Feature '{name}' is missing in {side}.")
    if y.shape[0] != x.shape[0] or y.shape[2] != x.shape[2]:
        raise ValueError("y does not have the correct shape.")
    return list(feature_names) if (have_both and feature_names) else []


def _chamfer_distance_single_direction(x, y, x_lengths, y_lengths, x_features, y_features, weights,
                                       point_reduction: Union[str, None], norm: int, abs_cosine: bool,
                                       feature_names: Union[list, None] = None):
    """(point term, feature terms or None) of the direction x -> y (reference: functions/chamfer.py:85-189)."""
    names = _direction_names(x, y, x_features, y_features, feature_names)
    N, P1, D = x.shape
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
            return _Terms(zero, {name: zero for name in names} if names else None)

    if _fused_direction_ok(x, y, x_features, y_features, names, bool(names), point_reduction):
        flat = [x_features[k] for k in names] + [y_features[k] for k in names]
        rows = _chamfer_direction.apply(x, y, x_lengths, y_lengths,
                                        None if weights is None else weights.to(torch.float32), norm,
                                        point_reduction == "mean", abs_cosine, *flat)
        return _Terms(rows[0], {k: rows[1 + i] for i, k in enumerate(names)} if names else None)
    return _direction_composed(x, y, x_lengths, y_lengths, x_features, y_features, names, weights, point_reduction,
                               norm, abs_cosine)


def _combine(fwd: _Terms, bwd: _Terms, point_reduction: Union[str, None]) -> _Terms:
    """Both directions into one result (reference: functions/chamfer.py:316-354): elementwise maximum for
    "max" (no feature loss), sum for "sum" / "mean", the (x, y) pair when there is no point reduction.  A feature
    the reverse direction lacks keeps the forward term (paired with None)."""
    if point_reduction == "max":
        return _Terms(torch.maximum(fwd.points, bwd.points), None)
    if point_reduction is not None:
        def join(a, b):
            return a if b is None else a + b
    else:
        def join(a, b):
            return (a, b)
    feats = None
    if fwd.feats is not None:
        other = bwd.feats or {}
        feats = {k: join(v, other.get(k)) for k, v in fwd.feats.items()}
    return _Terms(join(fwd.points, bwd.points), feats)


def _reduce_batch(terms: _Terms, weights, batch_reduction: Union[str, None]) -> _Terms:
    """Batch reduction of every term (reference: functions/chamfer.py:192-214): sum over clouds, and for
    "mean" a division by the weight sum (cloud count without weights; 1 when the weights sum to zero)."""
    if batch_reduction is None:
        return terms
    n_clouds = terms.points.shape[0]
    terms = terms.map(torch.sum)
    if batch_reduction == "sum":
        return terms
    if weights is None:
        denom = max(n_clouds, 1)
    else:
        total = weights.sum()
        denom = 1 if total == 0.0 else total
    return terms.map(lambda t: t / denom)


def _apply_batch_reduction(cham_x, cham_features_x, weights, batch_reduction: Union[str, None]):
    """Tuple form of `_reduce_batch` under the reference's helper name (functions/chamfer.py:192)."""
    return tuple(_reduce_batch(_Terms(cham_x, cham_features_x), weights, batch_reduction))


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
    if norm not in (1, 2):
        raise ValueError("Support for 1 or 2 norm.")
    if point_reduction == "max" and feature_names:
        raise ValueError('Features must be None if point_reduction is "max"')

    x, x_lengths, x_features = _handle_pointcloud_input(x, x_lengths, x_features)
    y, y_lengths, y_features = _handle_pointcloud_input(y, y_lengths, y_features)

    if not single_directional and weights is None and torch.is_tensor(x) and torch.is_tensor(y):
        names = _direction_names(x, y, x_features, y_features, feature_names)
        if (names == _direction_names(y, x, y_features, x_features, feature_names)
                and _fused_direction_ok(x, y, x_features, y_features, names, bool(names), point_reduction)
                and _fused_direction_ok(y, x, y_features, x_features, names, bool(names), point_reduction)):
            flat = [x_features[k] for k in names] + [y_features[k] for k in names]
            outs = _chamfer_pair.apply(x, y, x_lengths, y_lengths, norm, point_reduction == "mean", abs_cosine,
                                       batch_reduction, *flat)
            return outs[0], ({k: outs[1 + i] for i, k in enumerate(names)} if names else None)

    def direction(a, b, a_len, b_len, a_feat, b_feat):
        return _chamfer_distance_single_direction(a, b, a_len, b_len, a_feat, b_feat, weights, point_reduction, norm,
                                                  abs_cosine, feature_names)

    terms = direction(x, y, x_lengths, y_lengths, x_features, y_features)
    if not single_directional:
        terms = _combine(terms, direction(y, x, y_lengths, x_lengths, y_features, x_features), point_reduction)
    return tuple(_reduce_batch(terms, weights, batch_reduction))
