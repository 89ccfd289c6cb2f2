"""Host-side helpers shared by the functional wrappers: argument checks of point-set pairs, default
`lengths`, fp32 coercion for backward passes, and the nondeterminism alert of the atomic scatter kernels.

The `ValueError` texts are the reference's (pytorch3d_pointops/functions/knn.py:174-176,
functions/ball_query.py:120-123); everything else is this package's own plumbing.
"""
import warnings
from typing import Optional, Tuple

import torch

_LENGTHS_CACHE = {}


def full_lengths(n: int, p: int, device) -> torch.Tensor:
    """(n,) int64 tensor filled with p -- the default `lengths` (reference: functions/knn.py:184-187).
    The kernels only read it, so one tensor per (n, p, device) is kept and reused: it saves a fill launch per
    call, and passing the SAME tensor for both point sets lets the C ABI recognise a self-query (p1 is p2)."""
    key = (int(n), int(p), str(device))
    t = _LENGTHS_CACHE.get(key)
    if t is None:
        if len(_LENGTHS_CACHE) > 64:
            _LENGTHS_CACHE.clear()
        t = torch.full((n,), p, dtype=torch.int64, device=device)
        _LENGTHS_CACHE[key] = t
    return t


_MAX_CACHE = {}


def _forget_max(key):
    _MAX_CACHE.pop(key, None)


def lengths_max(lengths: torch.Tensor) -> int:
    """`int(lengths.max())`, remembered per tensor OBJECT and version counter: reading it back is a device-to-host
    copy that stalls the launch queue, and training loops validate the same lengths tensors call after call.
    An entry is keyed by id() and dropped by a weakref finalizer when its tensor dies; an in-place op bumps `_version`
    and invalidates it.  CAVEAT (tests/test_host_logic_cpu.py makes it explicit): a write that bypasses autograd's
    version counter -- `lengths.data[...] = v`, a raw-pointer write from another library or kernel -- is NOT seen; the
    kernels still clamp every length to the padded size, so the consequence of a stale maximum is a missing
    "length too large" error, never an out-of-bounds access."""
    import weakref
    key = id(lengths)
    hit = _MAX_CACHE.get(key)
    if hit is not None and hit[0]() is lengths and hit[1] == (lengths._version, lengths.data_ptr(), lengths.numel()):
        return hit[2]
    value = int(lengths.max()) if lengths.numel() else 0
    ref = weakref.ref(lengths)
    if hit is None:
        weakref.finalize(lengths, _forget_max, key)
    _MAX_CACHE[key] = (ref, (lengths._version, lengths.data_ptr(), lengths.numel()), value)
    return value


def batch_vector(value, n: int, device, name: str, message: str) -> torch.Tensor:
    """(n,) int64 device tensor from an int, a list or a tensor of per-cloud values; `message` is the ValueError text
    when the batch size does not match."""
    if isinstance(value, int):
        return torch.full((n,), value, dtype=torch.int64, device=device)
    t = torch.as_tensor(value, device=device) if not isinstance(value, torch.Tensor) else value.to(device)
    if t.dim() == 0 or t.shape[0] != n:
        raise ValueError(message)
    return t if t.dtype == torch.int64 else t.to(torch.int64)


def point_pair(p1: torch.Tensor, p2: torch.Tensor, lengths1: Optional[torch.Tensor],
               lengths2: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Validated, contiguous (p1, p2, lengths1, lengths2) of a query / reference pair of padded clouds."""
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    same = p1 is p2
    p1 = p1.contiguous()
    p2 = p1 if same else p2.contiguous()
    n = p1.shape[0]
    if lengths1 is None:
        lengths1 = full_lengths(n, p1.shape[1], p1.device)
    if lengths2 is None:
        lengths2 = full_lengths(n, p2.shape[1], p1.device)
    return p1, p2, lengths1, lengths2


def as_f32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.float32 else t.float()


def deterministic_requested() -> bool:
    """torch.use_deterministic_algorithms(True) is in force: the scatter sides of the backward passes then run through
    the inverted neighbour table (csrc/backward_det.hip) instead of fp32 scatter-adds."""
    return torch.are_deterministic_algorithms_enabled()


def alert_not_deterministic(caller: str) -> None:
    """What `at::globalContext().alertNotDeterministic(caller)` does in the reference's CUDA backward
    (csrc/knn/knn.cu:538): under `torch.use_deterministic_algorithms(True)` an op whose result depends on
    the order of fp32 atomic adds raises (or warns, with warn_only=True).  Only the FUSED chamfer kernels still
    call it, and only when they are invoked directly: `chamfer_distance` itself takes the composed path over the
    deterministic knn / gather backward passes when determinism is requested."""
    if torch.are_deterministic_algorithms_enabled():
        msg = (f"{caller} does not have a deterministic implementation, but you set "
               "'torch.use_deterministic_algorithms(True)'. You can turn off determinism just for this operation, "
               "or you can use the 'warn_only=True' option, if that's acceptable for your application.")
        if torch.is_deterministic_algorithms_warn_only_enabled():
            warnings.warn(msg, UserWarning, stacklevel=3)
        else:
            raise RuntimeError(msg)


def neighbor_backward(saved, norm: int, grad_dists: torch.Tensor, caller: str):
    """Shared backward of knn_points and ball_query: (grad_p1, grad_p2) from the neighbour table
    (reference: functions/knn.py:96-111, functions/ball_query.py:36-52).  idx == -1 entries are skipped
    by the kernel.  Under torch.use_deterministic_algorithms(True): the reproducible form, bit-equal to the
    reference's CPU backward."""
    from .. import _C

    p1, p2, lengths1, lengths2, idx = saved
    return _C.knn_points_backward(as_f32(p1), as_f32(p2), lengths1, lengths2, idx, norm, as_f32(grad_dists),
                                  deterministic=deterministic_requested())
