"""packed_to_padded / padded_to_packed -- same API as the reference's
functions/packed_to_padded.py (:15-62 _PackedToPadded, :65-103, :106-151
_PaddedToPacked, :154-198)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C


class _PackedToPadded(Function):
    @staticmethod
    def forward(ctx, inputs, first_idxs, max_size):
        if not (inputs.dim() == 2):
            raise ValueError("input can only be 2-dimensional.")
        if not (first_idxs.dim() == 1):
            raise ValueError("first_idxs can only be 1-dimensional.")
        if not (inputs.dtype == torch.float32):
            raise ValueError("input has to be of type torch.float32.")
        if not (first_idxs.dtype == torch.int64):
            raise ValueError("first_idxs has to be of type torch.int64.")
        if not isinstance(max_size, int):
            raise ValueError("max_size has to be int.")

        ctx.save_for_backward(first_idxs)
        ctx.num_inputs = int(inputs.shape[0])
        inputs, first_idxs = inputs.contiguous(), first_idxs.contiguous()
        return _C.packed_to_padded(inputs, first_idxs, max_size)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        first_idxs = ctx.saved_tensors[0]
        grad_input = _C.padded_to_packed(grad_output, first_idxs.contiguous(), ctx.num_inputs)
        return grad_input, None, None


def packed_to_padded(inputs: torch.Tensor, first_idxs: torch.LongTensor, max_size: int) -> torch.Tensor:
    """(F,) / (F, ...) packed -> (N, max_size) / (N, max_size, ...) zero-padded.

    Same contract as the reference (functions/packed_to_padded.py:65-103):
    ``first_idxs[i]`` is the first packed row of batch element i.
    """
    input_shape = inputs.shape
    n_dims = inputs.dim()
    if n_dims == 1:
        inputs = inputs.unsqueeze(1)
    else:
        inputs = inputs.reshape(input_shape[0], -1)
    inputs_padded = _PackedToPadded.apply(inputs, first_idxs, max_size)
    if n_dims == 1:
        return inputs_padded.squeeze(2)
    if n_dims == 2:
        return inputs_padded
    return inputs_padded.view(*inputs_padded.shape[:2], *input_shape[1:])


class _PaddedToPacked(Function):
    @staticmethod
    def forward(ctx, inputs, first_idxs, num_inputs):
        if not (inputs.dim() == 3):
            raise ValueError("input can only be 3-dimensional.")
        if not (first_idxs.dim() == 1):
            raise ValueError("first_idxs can only be 1-dimensional.")
        if not (inputs.dtype == torch.float32):
            raise ValueError("input has to be of type torch.float32.")
        if not (first_idxs.dtype == torch.int64):
            raise ValueError("first_idxs has to be of type torch.int64.")
        if not isinstance(num_inputs, int):
            raise ValueError("max_size has to be int.")

        ctx.save_for_backward(first_idxs)
        ctx.max_size = inputs.shape[1]
        inputs, first_idxs = inputs.contiguous(), first_idxs.contiguous()
        return _C.padded_to_packed(inputs, first_idxs, num_inputs)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        first_idxs = ctx.saved_tensors[0]
        grad_input = _C.packed_to_padded(grad_output, first_idxs.contiguous(), ctx.max_size)
        return grad_input, None, None


def padded_to_packed(
    inputs: torch.Tensor,
    first_idxs: torch.LongTensor,
    num_inputs: int,
    max_size_dim: int = 1,
) -> torch.Tensor:
    """(N, ..., max_size, ...) padded -> (F,) / (F, ...) packed.

    Same contract as the reference (functions/packed_to_padded.py:154-198).
    """
    n_dims = inputs.dim()
    inputs = inputs.movedim(max_size_dim, 1)
    input_shape = inputs.shape
    if n_dims == 2:
        inputs = inputs.unsqueeze(2)
    else:
        inputs = inputs.reshape(*input_shape[:2], -1)
    inputs_packed = _PaddedToPacked.apply(inputs, first_idxs, num_inputs)
    if n_dims == 2:
        return inputs_packed.squeeze(1)
    return inputs_packed.view(-1, *input_shape[2:])
