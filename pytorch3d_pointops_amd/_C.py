"""Operator boundary: the same callables as the reference's pybind11 module
``pytorch3d_pointops._C`` (reference: csrc/ext.cpp:15-27), implemented as ctypes
calls into ``lib/libpointops_amd.so`` (C ABI: include/pointops_amd.h).

PyTorch is plumbing only here: it owns device memory (outputs are fresh tensors on
the input device, like the reference's ``at::zeros`` / ``at::full``), the current
HIP stream and the device guard.  There is NO CPU implementation behind these
functions: CPU tensors raise ``RuntimeError`` (the mirror image of the reference's
"Not compiled with GPU support." -- csrc/knn/knn.h:74), and a missing shared
library raises at import of this module.
"""
import ctypes
import os

import torch  # must be imported first: loads the process-wide HIP runtime (libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
# POINTOPS_AMD_LIB: another build of the same library (tools/build_variant.py tuning experiments)
LIB_PATH = os.environ.get("POINTOPS_AMD_LIB") or os.path.join(_HERE, "lib", "libpointops_amd.so")

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/pointops_amd.h
_SIGNATURES = {
    "pointops_abi_version": (_int, []),
    "pointops_target_arch": (ctypes.c_char_p, []),
    "pointops_last_error": (ctypes.c_char_p, []),
    "pointops_knn_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64, _int]),
    "pointops_knn_points_idx": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _i64, _int,
                                       _vp, _vp, _vp, _sz, _vp]),
    "pointops_knn_uses_grid": (_int, [_i64, _i64, _i64, _i64, _i64, _int]),
    "pointops_knn_points_idx_reuse": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _i64, _int,
                                             _vp, _vp, _vp, _sz, _int, _vp]),
    "pointops_knn_check_version": (_int, [_int, _i64, _i64]),
    "pointops_knn_grid_fallback_counts": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "pointops_knn_grid_stats": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "pointops_knn_points_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                            _i64, _int, _vp, _vp, _vp]),
    "pointops_ball_query_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "pointops_ball_query": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _sz,
                                  _vp]),
    "pointops_fps_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "pointops_sample_farthest_points": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp,
                                               _sz, _vp]),
    "pointops_packed_to_padded": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "pointops_padded_to_packed": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "pointops_gather_neighbors": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "pointops_gather_neighbors_backward": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp,
                                                  _vp]),
    "pointops_backward_det_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "pointops_knn_points_backward_det": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp,
                                                _vp, _vp, _sz, _vp]),
    "pointops_gather_neighbors_backward_det": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "pointops_chamfer_reduce": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp]),
    "pointops_sample_pdf": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _f32, _vp]),
    "pointops_point_covariances": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "pointops_point_covariances_backward": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "pointops_chamfer_workspace_bytes": (_sz, [_i64, _i64]),
    "pointops_chamfer_forward": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _int,
                                        _int, _vp, _vp, _sz, _vp]),
    "pointops_chamfer_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int,
                                         _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "pointops_chamfer_backward_accumulate": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int,
                                                    _int, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "pointops_chamfer_pair_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _int]),
    "pointops_chamfer_pair_forward": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _vp,
                                             _int, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pointops_chamfer_pair_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int,
                                              _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m pytorch3d_pointops_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pointops_abi_version() != 1:
        raise ImportError("libpointops_amd.so ABI version mismatch")
    return lib


_lib = _load()


def exported_symbols():
    return sorted(_SIGNATURES)


def _check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {_lib.pointops_last_error().decode()}")


def _require_gpu(*tensors):
    dev = None
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "pytorch3d_pointops_amd is a GPU-only (MI355X / gfx950) implementation: got a CPU "
                "tensor and there is no CPU fallback."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("All tensors must be on the same GPU device")
    return dev


def _contig(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _f32c(t, name):
    """fp32 + contiguous view of a device tensor whose data pointer goes to a kernel."""
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float for {name}")
    return t.contiguous()


def _i64c(t, name):
    if t.dtype != torch.int64:
        raise RuntimeError(f"{name} must be int64")
    return t.contiguous()


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """Device guard for the launches: torch.cuda.device(dev) only when `dev` is not already current (its constructor,
    __enter__ and __exit__ cost ~4 us of a small call; the usual single-GPU process never needs the switch)."""
    return _NO_GUARD if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (as an int)."""
    if _raw_stream is not None:  # one C call instead of building a torch.cuda.Stream object
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------
# Grid reuse (opt-in): the cell grid of the exact search is an index over p2 -- bounding boxes, cell tables, the
# cell-sorted copy of the cloud, refined cells: 123 us of a 705 us call at B=32, N=M=65536, K=16 -- that a second
# query of the SAME target cloud does not have to rebuild (chamfer against a fixed ground truth, knn_points followed
# by further queries).  With the switch on, knn_points_idx keeps the workspaces of its last few grid calls and hands
# them back to the C ABI (pointops_knn_points_idx_reuse) when the target tensors are provably the ones it was built
# from: the same tensor OBJECTS (weak references), the same data pointers and the same autograd version counters --
# every in-place op of PyTorch bumps that counter.  It is OFF by default because one kind of write is invisible to
# it: `p2.data.copy_(...)` / a raw-pointer write from another library changes the bytes without the counter, and a
# stale grid then answers for the OLD points (the reference's stateless operator has no such failure mode).
#   pytorch3d_pointops_amd.set_grid_cache(True [, max_entries])      or      POINTOPS_GRID_CACHE=1
# ---------------------------------------------------------------------------
import collections
import weakref

_GRID_CACHE = collections.OrderedDict()
_GRID_CACHE_ON = os.environ.get("POINTOPS_GRID_CACHE", "0") not in ("", "0")
_GRID_CACHE_MAX = 2
grid_cache_stats = {"miss": 0, "points": 0, "both": 0}


def set_grid_cache(enabled: bool, max_entries: int = 2) -> None:
    """Switch the grid reuse of knn_points_idx on or off (off: the default; cached workspaces are dropped)."""
    global _GRID_CACHE_ON, _GRID_CACHE_MAX
    _GRID_CACHE_ON = bool(enabled)
    _GRID_CACHE_MAX = max(1, int(max_entries))
    if not enabled:
        _GRID_CACHE.clear()


_SCRATCH = {}


def _scratch(nbytes: int, dev):
    """Workspace for one call.  Small ones (<= 4 MiB: the sliced brute-force scans of small batches) come from a grow-only
    buffer per (device, stream): every user enqueues on that stream, in order, so the buffer can be handed out again at
    once and a small call saves an allocator round trip (~2 us).  Big ones are allocated per call."""
    if nbytes > (4 << 20) or torch.cuda.is_current_stream_capturing():
        # (inside a HIP-graph capture the buffer must belong to the graph's private pool: graphs.py)
        return torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    key = (dev.index, _stream())
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if len(_SCRATCH) > 32:
            _SCRATCH.clear()
        buf = torch.empty((max(nbytes, 1 << 16),), dtype=torch.uint8, device=dev)
        _SCRATCH[key] = buf
    return buf


def _sig(t):
    return (id(t), t._version, t.data_ptr(), tuple(t.shape))


def _grid_workspace(p1, p2, lengths1, lengths2, shape, ws_bytes, dev):
    """(workspace, reuse level) for a grid call: level 2 when both point sets are the cached call's, 1 when the
    target side is, 0 (a fresh workspace, remembered) otherwise."""
    key = (_sig(p2), _sig(lengths2), shape, _stream(), str(dev))
    hit = _GRID_CACHE.get(key)
    if hit is not None and hit["p2"]() is p2 and hit["l2"]() is lengths2 and hit["ws"].numel() == ws_bytes:
        _GRID_CACHE.move_to_end(key)
        q = (_sig(p1), _sig(lengths1))
        level = 2 if (hit["q"] == q and hit["p1"]() is p1 and hit["l1"]() is lengths1) else 1
        hit.update(q=q, p1=weakref.ref(p1), l1=weakref.ref(lengths1))
        grid_cache_stats["both" if level == 2 else "points"] += 1
        return hit["ws"], level
    grid_cache_stats["miss"] += 1
    for k in [k for k, v in _GRID_CACHE.items() if v["p2"]() is None]:
        del _GRID_CACHE[k]  # entries whose target tensor has died
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    _GRID_CACHE[key] = dict(ws=ws, p2=weakref.ref(p2), l2=weakref.ref(lengths2), q=(_sig(p1), _sig(lengths1)),
                            p1=weakref.ref(p1), l1=weakref.ref(lengths1))
    while len(_GRID_CACHE) > _GRID_CACHE_MAX:
        _GRID_CACHE.popitem(last=False)
    return ws, 0


# ---------------------------------------------------------------------------
# reference: csrc/knn/knn.h:59-80 -- returns (idx, dists), NOT (dists, idx)
# ---------------------------------------------------------------------------
def knn_points_idx(p1, p2, lengths1, lengths2, norm: int, K: int, version: int = -1):
    dev = _require_gpu(p1, p2, lengths1, lengths2)
    if p1.dtype != torch.float32 or p2.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float for p1/p2")  # CPU ref: same restriction
    if lengths1.dtype != torch.int64 or lengths2.dtype != torch.int64:
        raise RuntimeError("lengths1/lengths2 must be int64")
    p1, p2 = p1.contiguous(), p2.contiguous()  # reference CUDA path: knn.cu:373-376
    lengths1, lengths2 = lengths1.contiguous(), lengths2.contiguous()
    # (a self-query -- the same storage for both point sets and both lengths -- is recognised by the C ABI
    # from pointer equality and sorts the cloud once)
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    if p2.shape[0] != N or p2.shape[2] != D or lengths1.shape != (N,) or lengths2.shape != (N,):
        raise RuntimeError("knn_points_idx: inconsistent shapes")
    with _on(dev):
        idxs = torch.empty((N, P1, K), dtype=torch.int64, device=dev)
        dists = torch.empty((N, P1, K), dtype=torch.float32, device=dev)
        ws_bytes = _lib.pointops_knn_workspace_bytes(N, P1, P2, D, K, version)
        reuse = 0
        if _GRID_CACHE_ON and ws_bytes and _lib.pointops_knn_uses_grid(N, P1, P2, D, int(K), int(version)) \
                and not torch.cuda.is_current_stream_capturing():  # (a captured call must not bake a reuse level in)
            ws, reuse = _grid_workspace(p1, p2, lengths1, lengths2, (N, P1, P2, D, int(K), int(version)), ws_bytes, dev)
        else:
            ws = _scratch(ws_bytes, dev) if ws_bytes else None
        _check(
            _lib.pointops_knn_points_idx_reuse(p1.data_ptr(), p2.data_ptr(), lengths1.data_ptr(),
                                               lengths2.data_ptr(), N, P1, P2, D, int(norm), int(K),
                                               int(version), idxs.data_ptr(), dists.data_ptr(),
                                               ws.data_ptr() if ws is not None else None, ws_bytes, reuse,
                                               _stream()),
            "knn_points_idx",
        )
    return idxs, dists


def knn_grid_fallback_counts(p1, p2, lengths1, lengths2, norm: int, K: int):
    """Diagnostics: run the grid family (version 3) and return (idx, dists, counts) where
    counts[0, n] = queries of cloud n re-searched wave-per-query on a growing cell cube,
    counts[1, n] = queries that ended in the whole-cloud scan."""
    dev = _require_gpu(p1, p2, lengths1, lengths2)
    p1, p2 = p1.contiguous(), p2.contiguous()
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    if not knn_check_version(3, D, K):
        raise RuntimeError("grid family needs D <= 3 and K <= 128")
    with _on(dev):
        idxs = torch.empty((N, P1, K), dtype=torch.int64, device=dev)
        dists = torch.empty((N, P1, K), dtype=torch.float32, device=dev)
        ws_bytes = _lib.pointops_knn_workspace_bytes(N, P1, P2, D, K, 3)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        counts = torch.zeros((2, N), dtype=torch.int32, device=dev)
        _check(_lib.pointops_knn_points_idx(p1.data_ptr(), p2.data_ptr(), lengths1.data_ptr(),
                                            lengths2.data_ptr(), N, P1, P2, D, int(norm), int(K), 3,
                                            idxs.data_ptr(), dists.data_ptr(), ws.data_ptr(), ws_bytes,
                                            _stream()), "knn_points_idx")
        _check(_lib.pointops_knn_grid_fallback_counts(ws.data_ptr(), N, P1, P2, int(K), counts.data_ptr(),
                                                      _stream()), "knn_grid_fallback_counts")
    return idxs, dists, counts


def knn_grid_stats(p1, p2, lengths1, lengths2, norm: int, K: int):
    """Diagnostics: run the grid family and return (idx, dists, stats (N, 14) int32): cells per dimension (3),
    cell count, grid used, queries uncertified after the lane pass / the quad + box passes / sent to the whole-cloud
    scan, queries deferred to the box search, refined cells, bins of the point / query sort, crowded bins of the
    point / query sort."""
    dev = _require_gpu(p1, p2, lengths1, lengths2)
    p1 = p1.contiguous()
    p2 = p1 if p2 is p1 else p2.contiguous()
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    if not knn_check_version(3, D, K):
        raise RuntimeError("grid family needs D <= 3 and K <= 128")
    with _on(dev):
        idxs = torch.empty((N, P1, K), dtype=torch.int64, device=dev)
        dists = torch.empty((N, P1, K), dtype=torch.float32, device=dev)
        ws_bytes = _lib.pointops_knn_workspace_bytes(N, P1, P2, D, K, 3)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        stats = torch.zeros((N, 14), dtype=torch.int32, device=dev)
        _check(_lib.pointops_knn_points_idx(p1.data_ptr(), p2.data_ptr(), lengths1.data_ptr(),
                                            lengths2.data_ptr(), N, P1, P2, D, int(norm), int(K), 3,
                                            idxs.data_ptr(), dists.data_ptr(), ws.data_ptr(), ws_bytes,
                                            _stream()), "knn_points_idx")
        _check(_lib.pointops_knn_grid_stats(ws.data_ptr(), N, P1, P2, int(K), stats.data_ptr(), _stream()),
               "knn_grid_stats")
    return idxs, dists, stats


def knn_check_version(version: int, D: int, K: int) -> bool:
    """reference: csrc/knn/knn.h:161 (GPU builds only)."""
    return bool(_lib.pointops_knn_check_version(int(version), int(D), int(K)))


# reference: csrc/knn/knn.h:127-149
def knn_points_backward(p1, p2, lengths1, lengths2, idxs, norm: int, grad_dists, deterministic: bool = False):
    """`deterministic`: grad_p2 through the inverted neighbour table (csrc/backward_det.hip) -- reproducible and
    bit-equal to the reference's CPU kernel -- instead of fp32 scatter-adds (LDS tiles / device atomics)."""
    dev = _require_gpu(p1, p2, lengths1, lengths2, idxs, grad_dists)
    if p1.dtype != torch.float32 or p2.dtype != torch.float32 or grad_dists.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float")
    p1, p2 = p1.contiguous(), p2.contiguous()
    # the kernels read int64 through raw pointers: any other integer type would be read past its buffer (the
    # reference's accessor<int64_t, 1> raises: knn_cpu.cpp:84-88)
    lengths1, lengths2 = _i64c(lengths1, "lengths1"), _i64c(lengths2, "lengths2")
    idxs, grad_dists = _i64c(idxs, "idxs"), grad_dists.contiguous()
    if p1.dim() != 3 or p2.dim() != 3 or idxs.dim() != 3:
        raise RuntimeError("knn_points_backward: p1, p2 and idxs must be 3-dimensional")
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    K = idxs.shape[2]
    if (p2.shape[0] != N or p2.shape[2] != D or idxs.shape != (N, P1, K) or grad_dists.shape != (N, P1, K)
            or lengths1.shape != (N,) or lengths2.shape != (N,)):
        raise RuntimeError("knn_points_backward: inconsistent shapes")
    with _on(dev):
        grad_p1 = torch.empty((N, P1, D), dtype=torch.float32, device=dev)
        grad_p2 = torch.empty((N, P2, D), dtype=torch.float32, device=dev)
        if deterministic:
            ws_bytes = _lib.pointops_backward_det_workspace_bytes(N, P1, K, P2)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _check(
                _lib.pointops_knn_points_backward_det(p1.data_ptr(), p2.data_ptr(), lengths1.data_ptr(),
                                                      lengths2.data_ptr(), idxs.data_ptr(), grad_dists.data_ptr(),
                                                      N, P1, P2, D, K, int(norm), grad_p1.data_ptr(),
                                                      grad_p2.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
                "knn_points_backward(deterministic)",
            )
            return grad_p1, grad_p2
        _check(
            _lib.pointops_knn_points_backward(p1.data_ptr(), p2.data_ptr(), lengths1.data_ptr(),
                                              lengths2.data_ptr(), idxs.data_ptr(),
                                              grad_dists.data_ptr(), N, P1, P2, D, K, int(norm),
                                              grad_p1.data_ptr(), grad_p2.data_ptr(), _stream()),
            "knn_points_backward",
        )
    return grad_p1, grad_p2


# reference: csrc/ball_query/ball_query.h:62-93 -- returns (idx, dists)
def ball_query(p1, p2, lengths1, lengths2, K: int, radius: float):
    dev = _require_gpu(p1, p2, lengths1, lengths2)
    if p1.dtype != torch.float32 or p2.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float for p1/p2")
    p1, p2 = p1.contiguous(), p2.contiguous()  # ball_query.h:74-77
    lengths1, lengths2 = _i64c(lengths1, "lengths1"), _i64c(lengths2, "lengths2")  # (accessor<int64_t, 1> in the reference)
    if p1.dim() != 3 or p2.dim() != 3:
        raise RuntimeError("ball_query: p1 and p2 must be 3-dimensional")
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    if p2.shape[0] != N or p2.shape[2] != D or lengths1.shape != (N,) or lengths2.shape != (N,):
        raise RuntimeError("ball_query: inconsistent shapes")
    K = int(K)
    if K < 0:
        raise RuntimeError("ball_query: K must be non-negative")
    with _on(dev):
        idxs = torch.empty((N, P1, K), dtype=torch.int64, device=dev)
        dists = torch.empty((N, P1, K), dtype=torch.float32, device=dev)
        ws_bytes = _lib.pointops_ball_query_workspace_bytes(N, P1, P2, D, int(K))
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
        _check(
            _lib.pointops_ball_query(p1.data_ptr(), p2.data_ptr(), lengths1.data_ptr(),
                                     lengths2.data_ptr(), N, P1, P2, D, int(K), float(radius),
                                     idxs.data_ptr(), dists.data_ptr(),
                                     ws.data_ptr() if ws is not None else None, ws_bytes, _stream()),
            "ball_query",
        )
    return idxs, dists


# reference: csrc/sample_farthest_points/sample_farthest_points.h:55-76
def sample_farthest_points(points, lengths, K, start_idxs, max_K=None):
    dev = _require_gpu(points, lengths, K, start_idxs)
    if points.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float for points")
    points = points.contiguous()
    # int64 through raw pointers (the reference's accessor<int64_t, 1>: sample_farthest_points_cpu.cpp:33-36)
    lengths, K, start_idxs = _i64c(lengths, "lengths"), _i64c(K, "K"), _i64c(start_idxs, "start_idxs")
    if points.dim() != 3:
        raise RuntimeError("sample_farthest_points: points must be 3-dimensional")
    N, P, D = points.shape
    if lengths.shape != (N,) or K.shape != (N,) or start_idxs.shape != (N,):
        raise RuntimeError("sample_farthest_points: lengths, K and start_idxs must have shape (N,)")
    # host sync, as in the reference (sample_farthest_points.cu:132) -- unless the caller knows the maximum (an int or a
    # list K: no device-to-host read, and the call can be captured into a HIP graph)
    if max_K is None:
        max_K = int(K.max().item()) if N > 0 else 0
    max_K = int(max_K) if N > 0 else 0
    with _on(dev):
        idxs = torch.empty((N, max_K), dtype=torch.int64, device=dev)
        ws_bytes = _lib.pointops_fps_workspace_bytes(N, P, max_K)
        ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
        _check(
            _lib.pointops_sample_farthest_points(points.data_ptr(), lengths.data_ptr(), K.data_ptr(),
                                                 start_idxs.data_ptr(), N, P, D, max_K,
                                                 idxs.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
            "sample_farthest_points",
        )
    return idxs


# reference: csrc/packed_to_padded_tensor/packed_to_padded_tensor.h:78-94
def packed_to_padded(inputs_packed, first_idxs, max_size: int):
    dev = _require_gpu(inputs_packed, first_idxs)
    if inputs_packed.dim() != 2:
        raise RuntimeError("inputs_packed must be a 2-dimensional tensor")
    _contig(inputs_packed, "inputs_packed")
    _contig(first_idxs, "first_idxs")
    F, D = inputs_packed.shape
    B = first_idxs.shape[0]
    with _on(dev):
        out = torch.empty((B, max_size, D), dtype=torch.float32, device=dev)
        _check(
            _lib.pointops_packed_to_padded(inputs_packed.data_ptr(), first_idxs.data_ptr(), F, B,
                                           int(max_size), D, out.data_ptr(), _stream()),
            "packed_to_padded",
        )
    return out


# reference: csrc/packed_to_padded_tensor/packed_to_padded_tensor.h:97-113
def padded_to_packed(inputs_padded, first_idxs, num_inputs: int):
    dev = _require_gpu(inputs_padded, first_idxs)
    if inputs_padded.dim() != 3:
        raise RuntimeError("inputs_padded must be a 3-dimensional tensor")
    _contig(inputs_padded, "inputs_padded")
    _contig(first_idxs, "first_idxs")
    B, M, D = inputs_padded.shape
    with _on(dev):
        out = torch.empty((int(num_inputs), D), dtype=torch.float32, device=dev)
        _check(
            _lib.pointops_padded_to_packed(inputs_padded.data_ptr(), first_idxs.data_ptr(),
                                           int(num_inputs), B, M, D, out.data_ptr(), _stream()),
            "padded_to_packed",
        )
    return out


# reference: csrc/sample_pdf/sample_pdf.h:58-78 -- in place on `outputs`, returns None
def sample_pdf(bins, weights, outputs, eps: float):
    dev = _require_gpu(bins, weights, outputs)
    if bins.dtype != torch.float32 or weights.dtype != torch.float32 or outputs.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float")
    if not outputs.is_contiguous():
        raise RuntimeError("outputs must be contiguous")  # CHECK_CONTIGUOUS (sample_pdf.h:74)
    bins, weights = bins.contiguous(), weights.contiguous()
    batch, n_bins = weights.shape
    if bins.shape != (batch, n_bins + 1) or outputs.shape[0] != batch:
        raise RuntimeError("sample_pdf: inconsistent shapes")
    with _on(dev):
        _check(
            _lib.pointops_sample_pdf(bins.data_ptr(), weights.data_ptr(), outputs.data_ptr(), batch, n_bins,
                                     outputs.shape[1], float(eps), _stream()),
            "sample_pdf",
        )


# --- fused device half of get_point_covariances (functions/utils.py:111-153) -------
POINT_COVARIANCES_MAX_D = 8


def point_covariances(knn):
    """knn (N,P,K,D) fp32 -> cov (N,P,D,D)."""
    dev = _require_gpu(knn)
    knn = knn.contiguous()
    N, P, K, D = knn.shape
    with _on(dev):
        cov = torch.empty((N, P, D, D), dtype=torch.float32, device=dev)
        _check(_lib.pointops_point_covariances(knn.data_ptr(), N, P, K, D, cov.data_ptr(), _stream()),
               "point_covariances")
    return cov


def point_covariances_backward(knn, grad_cov):
    dev = _require_gpu(knn, grad_cov)
    knn, grad_cov = knn.contiguous(), grad_cov.contiguous()
    N, P, K, D = knn.shape
    with _on(dev):
        grad_knn = torch.empty_like(knn)
        _check(_lib.pointops_point_covariances_backward(knn.data_ptr(), grad_cov.data_ptr(), N, P, K, D,
                                                        grad_knn.data_ptr(), _stream()),
               "point_covariances_backward")
    return grad_knn


# --- device halves of knn_gather / masked_gather (functions/knn.py:200-250) -------
def gather_neighbors(x, idx, lengths=None):
    dev = _require_gpu(x, idx) if lengths is None else _require_gpu(x, idx, lengths)
    x, idx = _f32c(x, "x"), _i64c(idx, "idx")
    lengths = _i64c(lengths, "lengths") if lengths is not None else None
    N, M, U = x.shape
    _, L, K = idx.shape
    with _on(dev):
        out = torch.empty((N, L, K, U), dtype=torch.float32, device=dev)
        _check(
            _lib.pointops_gather_neighbors(x.data_ptr(), idx.data_ptr(),
                                           lengths.data_ptr() if lengths is not None else None,
                                           N, M, U, L, K, out.data_ptr(), _stream()),
            "gather_neighbors",
        )
    return out


def gather_neighbors_backward(grad_out, idx, lengths, M: int, deterministic: bool = False):
    dev = _require_gpu(grad_out, idx) if lengths is None else _require_gpu(grad_out, idx, lengths)
    grad_out, idx = _f32c(grad_out, "grad_out"), _i64c(idx, "idx")
    lengths = _i64c(lengths, "lengths") if lengths is not None else None
    N, L, K, U = grad_out.shape
    with _on(dev):
        grad_x = torch.empty((N, M, U), dtype=torch.float32, device=dev)
        if deterministic:  # inverted neighbour table: every row of x sums its addends in table order
            ws_bytes = _lib.pointops_backward_det_workspace_bytes(N, L, K, M)
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _check(
                _lib.pointops_gather_neighbors_backward_det(
                    grad_out.data_ptr(), idx.data_ptr(), lengths.data_ptr() if lengths is not None else None,
                    N, M, U, L, K, grad_x.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
                "gather_neighbors_backward(deterministic)",
            )
            return grad_x
        _check(
            _lib.pointops_gather_neighbors_backward(
                grad_out.data_ptr(), idx.data_ptr(),
                lengths.data_ptr() if lengths is not None else None, N, M, U, L, K,
                grad_x.data_ptr(), _stream()),
            "gather_neighbors_backward",
        )
    return grad_x


def chamfer_reduce(dists, lengths, weights, mean: bool):
    """dists (N,P) fp32 -> (N,) per-cloud masked sum [* weights] [/ max(len,1)]."""
    dev = _require_gpu(dists, lengths) if weights is None else _require_gpu(dists, lengths, weights)
    dists, lengths = _f32c(dists, "dists"), _i64c(lengths, "lengths")
    weights = _f32c(weights, "weights") if weights is not None else None
    N, P = dists.shape
    if lengths.shape != (N,) or (weights is not None and weights.shape != (N,)):
        raise RuntimeError("chamfer_reduce: lengths / weights must have shape (N,)")
    with _on(dev):
        out = torch.empty((N,), dtype=torch.float32, device=dev)
        _check(
            _lib.pointops_chamfer_reduce(dists.data_ptr(), lengths.data_ptr(),
                                         weights.data_ptr() if weights is not None else None,
                                         N, P, int(bool(mean)), out.data_ptr(), _stream()),
            "chamfer_reduce",
        )
    return out


# --- fused single-direction chamfer terms (functions/chamfer.py:135-185) --------------------
CHAMFER_MAX_FEATURES = 4
CHAMFER_MAX_CHANNELS = 16


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _check_chamfer_shapes(N, P1, P2, idx, x_lengths, y_lengths, weights, x_feats, y_feats):
    if len(x_feats) != len(y_feats) or len(x_feats) > CHAMFER_MAX_FEATURES:
        raise RuntimeError(f"chamfer: at most {CHAMFER_MAX_FEATURES} feature pairs")
    if idx.shape != (N, P1) or x_lengths.shape != (N,) or y_lengths.shape != (N,):
        raise RuntimeError("chamfer: idx must be (N, P1) and the lengths (N,)")
    if weights is not None and weights.shape != (N,):
        raise RuntimeError("chamfer: weights must have shape (N,)")
    for a, b in zip(x_feats, y_feats):
        if a.dim() != 3 or b.dim() != 3 or a.shape[:2] != (N, P1) or b.shape[:2] != (N, P2) \
                or a.shape[2] != b.shape[2] or not 1 <= a.shape[2] <= CHAMFER_MAX_CHANNELS:
            raise RuntimeError("chamfer: features must be (N, P1, C) / (N, P2, C) with 1 <= C <= "
                               f"{CHAMFER_MAX_CHANNELS}")


def chamfer_forward(dists, idx, x_lengths, y_lengths, weights, x_feats, y_feats, abs_cosine: bool, mean: bool):
    """dists (N,P1) fp32 and idx (N,P1) int64 of the K=1 search -> out (1+F, N): row 0 the point term,
    row 1+f the cosine term of feature f, each already weighted and (for "mean") length-normalised."""
    opt = [weights] if weights is not None else []
    dev = _require_gpu(dists, idx, x_lengths, y_lengths, *opt, *x_feats, *y_feats)
    dists, idx = _f32c(dists, "dists"), _i64c(idx, "idx")
    x_lengths, y_lengths = _i64c(x_lengths, "x_lengths"), _i64c(y_lengths, "y_lengths")
    weights = _f32c(weights, "weights") if weights is not None else None
    x_feats = [_f32c(t, "x_feats") for t in x_feats]
    y_feats = [_f32c(t, "y_feats") for t in y_feats]
    N, P1 = dists.shape
    F = len(x_feats)
    P2 = y_feats[0].shape[1] if F else 0
    _check_chamfer_shapes(N, P1, P2, idx, x_lengths, y_lengths, weights, x_feats, y_feats)
    C = (ctypes.c_int64 * max(F, 1))(*[int(t.shape[2]) for t in x_feats])
    with _on(dev):
        out = torch.empty((1 + F, N), dtype=torch.float32, device=dev)
        ws_bytes = _lib.pointops_chamfer_workspace_bytes(N, P1)
        ws = torch.empty((max(ws_bytes, 4),), dtype=torch.uint8, device=dev)
        _check(
            _lib.pointops_chamfer_forward(dists.data_ptr(), idx.data_ptr(), x_lengths.data_ptr(),
                                          y_lengths.data_ptr(),
                                          weights.data_ptr() if weights is not None else None, N, P1, P2, F,
                                          _ptr_array(x_feats), _ptr_array(y_feats), C, int(bool(abs_cosine)),
                                          int(bool(mean)), out.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
            "chamfer_forward",
        )
    return out


def chamfer_backward(x, y, idx, x_lengths, y_lengths, weights, grad_out, norm: int, x_feats, y_feats,
                     abs_cosine: bool, mean: bool, into=None):
    """Closed-form gradients of chamfer_forward's outputs: returns (grad_x, grad_y, [grad_x_feat], [grad_y_feat]).
    `into` = (grad_x, grad_y, [grad_x_feat], [grad_y_feat]) buffers that already hold gradients (the other direction's,
    roles swapped): the gradients are ADDED to them (pointops_chamfer_backward_accumulate) and they are returned."""
    opt = [weights] if weights is not None else []
    dev = _require_gpu(x, y, idx, grad_out, x_lengths, y_lengths, *opt, *x_feats, *y_feats)
    x, y, grad_out = _f32c(x, "x"), _f32c(y, "y"), _f32c(grad_out, "grad_out")
    idx = _i64c(idx, "idx")
    x_lengths, y_lengths = _i64c(x_lengths, "x_lengths"), _i64c(y_lengths, "y_lengths")
    weights = _f32c(weights, "weights") if weights is not None else None
    x_feats = [_f32c(t, "x_feats") for t in x_feats]
    y_feats = [_f32c(t, "y_feats") for t in y_feats]
    N, P1, D = x.shape
    P2 = y.shape[1]
    F = len(x_feats)
    if y.shape[0] != N or y.shape[2] != D or grad_out.shape != (1 + F, N):
        raise RuntimeError("chamfer_backward: inconsistent shapes")
    _check_chamfer_shapes(N, P1, P2, idx, x_lengths, y_lengths, weights, x_feats, y_feats)
    C = (ctypes.c_int64 * max(F, 1))(*[int(t.shape[2]) for t in x_feats])
    with _on(dev):
        if into is None:
            grad_x = torch.empty_like(x)
            grad_y = torch.empty_like(y)
            gxf = [torch.empty_like(t) for t in x_feats]
            gyf = [torch.empty_like(t) for t in y_feats]
            entry = _lib.pointops_chamfer_backward
        else:
            grad_x, grad_y, gxf, gyf = into
            gxf, gyf = list(gxf), list(gyf)
            for g, like in zip([grad_x, grad_y, *gxf, *gyf], [x, y, *x_feats, *y_feats]):
                if (g.shape != like.shape or g.dtype != torch.float32 or not g.is_contiguous()
                        or g.device != like.device):
                    raise RuntimeError("chamfer_backward: `into` buffers must match the inputs (fp32, contiguous)")
            entry = _lib.pointops_chamfer_backward_accumulate
        _check(
            entry(x.data_ptr(), y.data_ptr(), idx.data_ptr(), x_lengths.data_ptr(), y_lengths.data_ptr(),
                  weights.data_ptr() if weights is not None else None, grad_out.data_ptr(), N, P1, P2, D, int(norm),
                  F, _ptr_array(x_feats), _ptr_array(y_feats), C, int(bool(abs_cosine)), int(bool(mean)),
                  grad_x.data_ptr(), grad_y.data_ptr(), _ptr_array(gxf), _ptr_array(gyf), _stream()),
            "chamfer_backward",
        )
    return grad_x, grad_y, gxf, gyf


_BATCH_REDUCTION = {None: 0, "mean": 1, "sum": 2}


def grid_cache_enabled() -> bool:
    return _GRID_CACHE_ON


def chamfer_pair_forward(x, y, x_lengths, y_lengths, norm: int, x_feats, y_feats, abs_cosine: bool, mean: bool,
                         batch_reduction):
    """Both directions of an unweighted chamfer distance in ONE native call (pointops_chamfer_pair_forward):
    returns (outs, idx_xy (N,P1), idx_yx (N,P2)); outs = 1+F tensors, () after a batch reduction, (N,) without one."""
    dev = _require_gpu(x, y, x_lengths, y_lengths, *x_feats, *y_feats)
    x, y = _f32c(x, "x"), _f32c(y, "y")
    x_lengths, y_lengths = _i64c(x_lengths, "x_lengths"), _i64c(y_lengths, "y_lengths")
    x_feats = [_f32c(t, "x_feats") for t in x_feats]
    y_feats = [_f32c(t, "y_feats") for t in y_feats]
    if norm not in (1, 2):
        raise ValueError("Support for 1 or 2 norm.")
    N, P1, D = x.shape
    P2 = y.shape[1]
    F = len(x_feats)
    if y.shape[0] != N or y.shape[2] != D:
        raise RuntimeError("chamfer_pair_forward: inconsistent shapes")
    for a, b in zip(x_feats, y_feats):  # (the index shapes are ours; the feature checks are the single direction's)
        if a.dim() != 3 or b.dim() != 3 or a.shape[:2] != (N, P1) or b.shape[:2] != (N, P2) \
                or a.shape[2] != b.shape[2] or not 1 <= a.shape[2] <= CHAMFER_MAX_CHANNELS:
            raise RuntimeError("chamfer: features must be (N, P1, C) / (N, P2, C) with 1 <= C <= "
                               f"{CHAMFER_MAX_CHANNELS}")
    if len(x_feats) != len(y_feats) or F > CHAMFER_MAX_FEATURES or x_lengths.shape != (N,) or y_lengths.shape != (N,):
        raise RuntimeError(f"chamfer: at most {CHAMFER_MAX_FEATURES} feature pairs, lengths of shape (N,)")
    C = (ctypes.c_int64 * max(F, 1))(*[int(t.shape[2]) for t in x_feats])
    red = _BATCH_REDUCTION[batch_reduction]
    with _on(dev):
        idx_xy = torch.empty((N, P1), dtype=torch.int64, device=dev)
        idx_yx = torch.empty((N, P2), dtype=torch.int64, device=dev)
        outs = [torch.empty(() if red else (N,), dtype=torch.float32, device=dev) for _ in range(1 + F)]
        ws_bytes = _lib.pointops_chamfer_pair_workspace_bytes(N, P1, P2, D, F)
        ws = _scratch(ws_bytes, dev) if ws_bytes else None
        _check(
            _lib.pointops_chamfer_pair_forward(x.data_ptr(), y.data_ptr(), x_lengths.data_ptr(), y_lengths.data_ptr(), N,
                                               P1, P2, D, int(norm), F, _ptr_array(x_feats), _ptr_array(y_feats), C,
                                               int(bool(abs_cosine)), int(bool(mean)), red, idx_xy.data_ptr(),
                                               idx_yx.data_ptr(), _ptr_array(outs),
                                               ws.data_ptr() if ws is not None else None, ws_bytes, _stream()),
            "chamfer_pair_forward",
        )
    return outs, idx_xy, idx_yx


def chamfer_pair_backward(x, y, idx_xy, idx_yx, x_lengths, y_lengths, grads, norm: int, x_feats, y_feats,
                          abs_cosine: bool, mean: bool, batch_reduction):
    """Gradients of chamfer_pair_forward's 1+F outputs (`grads`: one tensor or None per output) in ONE native call:
    returns (grad_x, grad_y, [grad_x_feat], [grad_y_feat])."""
    live = [g for g in grads if g is not None]
    dev = _require_gpu(x, y, idx_xy, idx_yx, x_lengths, y_lengths, *live, *x_feats, *y_feats)
    N, P1, D = x.shape
    P2 = y.shape[1]
    F = len(x_feats)
    red = _BATCH_REDUCTION[batch_reduction]
    want = () if red else (N,)
    if len(grads) != 1 + F or any(g.shape != want for g in live):
        raise RuntimeError("chamfer_pair_backward: one gradient per output, () after a batch reduction, (N,) without")
    live = [_f32c(g, "grad") for g in live]
    it = iter(live)
    garr = (ctypes.c_void_p * (1 + F))(*[None if g is None else next(it).data_ptr() for g in grads])
    C = (ctypes.c_int64 * max(F, 1))(*[int(t.shape[2]) for t in x_feats])
    with _on(dev):
        grad_x = torch.empty_like(x)
        grad_y = torch.empty_like(y)
        gxf = [torch.empty_like(t) for t in x_feats]
        gyf = [torch.empty_like(t) for t in y_feats]
        ws_bytes = 4 * (1 + F) * N
        ws = _scratch(ws_bytes, dev)
        _check(
            _lib.pointops_chamfer_pair_backward(x.data_ptr(), y.data_ptr(), idx_xy.data_ptr(), idx_yx.data_ptr(),
                                                x_lengths.data_ptr(), y_lengths.data_ptr(), garr, N, P1, P2, D,
                                                int(norm), F, _ptr_array(x_feats), _ptr_array(y_feats), C,
                                                int(bool(abs_cosine)), int(bool(mean)), red, grad_x.data_ptr(),
                                                grad_y.data_ptr(), _ptr_array(gxf), _ptr_array(gyf), ws.data_ptr(),
                                                ws_bytes, _stream()),
            "chamfer_pair_backward",
        )
    return grad_x, grad_y, gxf, gyf
