"""packed_to_padded / padded_to_packed -- the API of the reference's functions/packed_to_padded.py
(:65-103, :154-198) on the HIP copy kernels (csrc/packed_padded.hip).

The two conversions are each other's adjoint, so ONE autograd node serves both: it runs the copy named by
`to_padded` forward and the opposite copy backward (reference: two mirrored Function classes, :15-62 and
:106-151).  Argument errors keep the reference's texts (:38-47, :130-139).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C


def _ragged_copy(data, first_idxs, size: int, to_padded: bool):
    return (_C.packed_to_padded if to_padded else _C.padded_to_packed)(data.contiguous(), first_idxs.contiguous(), size)


class _RaggedCopyFn(Function):
    """to_padded: (F, D) -> (N, size, D);  otherwise (N, M, D) -> (size, D)."""

    @staticmethod
    def forward(ctx, data, first_idxs, size, to_padded):
        problems = (
            (data.dim() != (2 if to_padded else 3), f"input can only be {2 if to_padded else 3}-dimensional."),
            (first_idxs.dim() != 1, "first_idxs can only be 1-dimensional."),
            (data.dtype != torch.float32, "input has to be of type torch.float32."),
            (first_idxs.dtype != torch.int64, "first_idxs has to be of type torch.int64."),
            (not isinstance(size, int), "max_size has to be int."),
        )
        for bad, text in problems:
            if bad:
                raise ValueError(text)
        ctx.to_padded = to_padded
        ctx.adjoint_size = int(data.shape[0] if to_padded else data.shape[1])  # F, or the padded length M
        ctx.save_for_backward(first_idxs)
        return _ragged_copy(data, first_idxs, size, to_padded)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        (first_idxs,) = ctx.saved_tensors
        return _ragged_copy(grad, first_idxs, ctx.adjoint_size, not ctx.to_padded), None, None, None


def packed_to_padded(inputs: torch.Tensor, first_idxs: torch.LongTensor, max_size: int) -> torch.Tensor:
    """Packed (F,) or (F, ...) -> zero-padded (N, max_size) or (N, max_size, ...); batch element i owns the
    packed rows from first_idxs[i] up to first_idxs[i+1] (the last one up to F).  Differentiable."""
    trailing = tuple(inputs.shape[1:])
    flat = inputs.reshape(inputs.shape[0], -1) if trailing else inputs[:, None]
    if torch.compiler.is_compiling():
        padded = torch.ops.pointops_amd.packed_to_padded(flat.contiguous(), first_idxs.contiguous(), max_size)
    else:
        padded = _RaggedCopyFn.apply(flat, first_idxs, max_size, True)
    return padded.reshape(padded.shape[0], padded.shape[1], *trailing)


def padded_to_packed(
    inputs: torch.Tensor,
    first_idxs: torch.LongTensor,
    num_inputs: int,
    max_size_dim: int = 1,
) -> torch.Tensor:
    """Padded (N, max_size) or (N, ..., max_size, ...) with the padded dimension at `max_size_dim` ->
    packed (num_inputs,) or (num_inputs, ...).  Differentiable."""
    moved = inputs.movedim(max_size_dim, 1)
    trailing = tuple(moved.shape[2:])
    flat = moved.reshape(moved.shape[0], moved.shape[1], -1) if trailing else moved[:, :, None]
    if torch.compiler.is_compiling():
        packed = torch.ops.pointops_amd.padded_to_packed(flat.contiguous(), first_idxs.contiguous(), num_inputs)
    else:
        packed = _RaggedCopyFn.apply(flat, first_idxs, num_inputs, False)
    return packed.reshape(packed.shape[0], *trailing)
