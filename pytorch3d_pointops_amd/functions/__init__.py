"""Functional API -- same re-exports as the reference's functions/__init__.py:9-17
(chamfer_distance and sample_pdf are imported by module path there too)."""
from .. import ops as _ops  # noqa: F401  registers torch.ops.pointops_amd.* (torch.compile traces through them)
from .ball_query import ball_query
from .knn import knn_gather, knn_points
from .packed_to_padded import packed_to_padded, padded_to_packed
from .sample_farthest_points import sample_farthest_points
from .utils import get_point_covariances, masked_gather, wmean

__all__ = [k for k in globals().keys() if not k.startswith("_")]
