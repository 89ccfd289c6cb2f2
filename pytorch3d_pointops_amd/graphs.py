"""HIP-graph capture of the hot path: one graph launch per step instead of one host launch per kernel.

A call of this package is a SEQUENCE of small dependent launches -- a grid-searched `knn_points` enqueues about a dozen
(bounding boxes, the two-level counting sort, the lane / quad / box / wave searches), a bidirectional
`chamfer_distance` with its backward about forty -- and below a few thousand points per cloud, or one call at a time,
the host's enqueue time is what a training step waits for (cfg4 chamfer 0.57 ms back to back,
0.69 ms one call at a time; B=4, N=2048 forward + backward 76 us of kernels inside 133-248 us).  The reference has no
answer to that on its CUDA path either; on MI355X the idiomatic one is a HIP graph: capture the sequence once, replay
it with one `hipGraphLaunch`.

Every operator of the package can be captured because the C ABI never synchronises, never allocates and takes every
data-dependent decision (grid or scan per cloud, which fallback pass a query needs) ON THE DEVICE; the chamfer's
reverse search forks to its side stream and joins back through events, which a capture follows.  The host side keeps
three promises while `torch.cuda.is_current_stream_capturing()`: workspaces come from the graph's private pool
(`_C._scratch`), the opt-in grid reuse is bypassed (a captured reuse level would be replayed blindly), and the lengths
validation's device-to-host read has happened during the warm-up calls (`_common.lengths_max` remembers it).

    step = capture(lambda x, y: chamfer_distance(x, y)[0], (x, y), backward=True)
    loss, (gx, gy) = step(new_x, new_y)      # copies into the static inputs, replays, returns the static results
    loss, (gx, gy) = step()                  # the caller has updated x / y in place

What a replay does NOT do is the Python-side argument validation of the captured call (shapes are frozen by the
capture; a `lengths` entry above the padded size is clamped by the kernels instead of raising).
"""
from typing import Callable, Optional, Sequence, Tuple

import torch


def _flatten(out):
    if isinstance(out, torch.Tensor):
        return (out,), True
    return tuple(out), False


class GraphedCall:
    """`fn(*inputs)` -- any composition of this package's operators and torch ops on `inputs`' device -- captured into
    a HIP graph.  `inputs` become the graph's static input buffers (the caller's own tensors: nothing is cloned).
    `fn` returns a tensor or a tuple of tensors; with `backward=True` its FIRST output must be a scalar and the
    gradients of that scalar with respect to every input that requires grad are part of the graph."""

    def __init__(self, fn: Callable, inputs: Sequence[torch.Tensor], backward: bool = False, warmup: int = 3):
        inputs = tuple(inputs)
        if not inputs or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in inputs):
            raise RuntimeError("GraphedCall: inputs must be GPU tensors (there is no CPU path)")
        self.inputs = inputs
        self._wrt = tuple(t for t in inputs if t.requires_grad) if backward else ()
        if backward and not self._wrt:
            raise RuntimeError("GraphedCall(backward=True): no input requires grad")

        def run():
            out, single = _flatten(fn(*self.inputs))
            grads = torch.autograd.grad(out[0], self._wrt) if self._wrt else ()
            return tuple(o.detach() for o in out), single, tuple(grads)

        dev = inputs[0].device
        with torch.cuda.device(dev):
            # warm-up on a side stream (torch's capture protocol): first-use work -- the library's side stream and
            # events, kernel attributes, cached default lengths, the lengths validation -- happens here, not under capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, int(warmup))):
                    run()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outputs, self._single, self.grads = run()
        # tensors the captured launches may point at although no Python object of the capture owns them
        from . import _C
        from .functions import _common
        self._keepalive = (list(_common._LENGTHS_CACHE.values()), list(_C._SCRATCH.values()))

    def replay(self) -> None:
        self.graph.replay()

    def __call__(self, *new_inputs: torch.Tensor):
        if new_inputs:
            if len(new_inputs) != len(self.inputs):
                raise RuntimeError(f"GraphedCall: expected {len(self.inputs)} inputs, got {len(new_inputs)}")
            with torch.no_grad():
                for dst, src in zip(self.inputs, new_inputs):
                    if src is not dst:
                        if src.shape != dst.shape or src.dtype != dst.dtype:
                            raise RuntimeError("GraphedCall: shapes and dtypes are frozen by the capture "
                                               f"({tuple(dst.shape)} {dst.dtype} vs {tuple(src.shape)} {src.dtype})")
                        dst.copy_(src)
        self.graph.replay()
        out = self.outputs[0] if self._single else self.outputs
        return (out, self.grads) if self._wrt else out


def capture(fn: Callable, inputs: Sequence[torch.Tensor], backward: bool = False, warmup: int = 3) -> GraphedCall:
    """Capture `fn(*inputs)` (and, with `backward=True`, the gradients of its first output) into a HIP graph."""
    return GraphedCall(fn, inputs, backward=backward, warmup=warmup)
