"""pytorch3d_pointops_amd -- MI355X-native (gfx950) batched point-cloud neighbour ops.

Drop-in for the hot path of ``pytorch3d_pointops`` (knn_points, knn_gather,
ball_query, sample_farthest_points, chamfer_distance, packed_to_padded /
padded_to_packed) behind the same ``functions.*`` Python API; the device work is
done by hand-written HIP kernels in ``lib/libpointops_amd.so`` reached through a
C ABI (``include/pointops_amd.h``).  There is no CPU fallback: a missing library
or a CPU tensor raises.
"""
__version__ = "0.1.0"

# `pytorch3d_pointops_amd.build` must stay importable before the library exists, so nothing is imported here:
# `functions` (and `structures`, which uses it) load `_C` -- the ctypes boundary, raising when the .so is
# missing -- and `ops`, which registers torch.ops.pointops_amd.*.


def set_grid_cache(enabled: bool, max_entries: int = 2) -> None:
    """Opt in to (or out of) the reuse of the exact search's cell grid between calls that query the same, unmodified
    target tensor -- see `_C.set_grid_cache` for what "unmodified" can and cannot see."""
    from . import _C

    _C.set_grid_cache(enabled, max_entries)
