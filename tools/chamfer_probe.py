#!/usr/bin/env python3
"""cfg4-shape chamfer fwd+bwd: time per call of a back-to-back loop (host and GPU overlapped) vs one call at a time
(host launch latency exposed); run under rocprofv3 --kernel-trace --stats for the kernel-time sum."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import synth  # noqa: E402
from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance  # noqa: E402

dev = torch.device("cuda:0")
Bq = 8
l1 = synth.randint(41, 20000, 200000, (Bq,))
l2 = synth.randint(42, 20000, 200000, (Bq,))
P1, P2 = int(l1.max()), int(l2.max())
x = torch.from_numpy(synth.uniform_f32(43, (Bq, P1, 3))).to(dev).requires_grad_(True)
y = torch.from_numpy(synth.uniform_f32(44, (Bq, P2, 3))).to(dev).requires_grad_(True)
xn = torch.from_numpy(synth.unit_normals(45, (Bq, P1, 3))).to(dev).requires_grad_(True)
yn = torch.from_numpy(synth.unit_normals(46, (Bq, P2, 3))).to(dev).requires_grad_(True)
xl, yl = torch.from_numpy(l1).to(dev), torch.from_numpy(l2).to(dev)


FRESH = "fresh" in sys.argv  # .grad = None before every call, as an optimizer's zero_grad(set_to_none=True) does:
                              # the gradients we return BECOME .grad; without it autograd adds them into the old ones
                              # (four elementwise adds over (N, P, 3) tensors, ~44 us per call at this size)


def fb():
    if FRESH:
        x.grad = y.grad = xn.grad = yn.grad = None
    loss, lf = chamfer_distance(x, y, x_lengths=xl, y_lengths=yl, x_features={"normals": xn},
                                y_features={"normals": yn}, feature_names=["normals"])
    (loss + lf["normals"]).backward()


iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    fb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    fb()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"back-to-back: {(t2 - t0) / iters * 1e3:.3f} ms per call (host enqueue {(t1 - t0) / iters * 1e3:.3f} ms per call)")
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    fb()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - a) * 1e3)
print(f"one at a time: median {np.median(ts):.3f} ms")
if "profile" in sys.argv:
    import cProfile
    import io
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        fb()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])
