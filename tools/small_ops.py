#!/usr/bin/env python3
"""Latency of every hot-path function at the reference's own example sizes (B=2, N=1024 .. 4096): back to back and one
at a time (synchronised after every call), through the public functions."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import synth  # noqa: E402
from pytorch3d_pointops_amd.functions import ball_query, knn_gather, knn_points, sample_farthest_points  # noqa: E402
from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance  # noqa: E402

dev = torch.device("cuda:0")


def timed(f, n=1000, sync=False):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if sync:
        for _ in range(n):
            f()
            torch.cuda.synchronize()
    else:
        for _ in range(n):
            f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for (B, N) in ((2, 1024), (2, 4096), (8, 2048)):
    a = torch.from_numpy(synth.uniform_f32(1, (B, N, 3))).to(dev)
    b = torch.from_numpy(synth.uniform_f32(2, (B, N, 3))).to(dev)
    ag = a.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True)
    idx = knn_points(a, b, K=8).idx

    def chamfer_fb():
        loss, _ = chamfer_distance(ag, bg)
        loss.backward()

    def knn_fb():
        r = knn_points(ag, bg, K=8)
        r.dists.sum().backward()

    rows = [("knn_points K=8", lambda: knn_points(a, b, K=8)), ("knn_points K=1", lambda: knn_points(a, b, K=1)),
            ("knn_points K=8 fwd+bwd", knn_fb),
            ("knn_gather", lambda: knn_gather(b, idx)),
            ("ball_query K=32 r=0.2", lambda: ball_query(a, b, K=32, radius=0.2)),
            ("ball_query K=16 r=0.05", lambda: ball_query(a, b, K=16, radius=0.05)),
            ("sample_farthest_points K=128", lambda: sample_farthest_points(a, K=128)),
            ("chamfer fwd", lambda: chamfer_distance(a, b)), ("chamfer fwd+bwd", chamfer_fb)]
    for name, f in rows:
        print("B=%d N=%-5d %-30s back to back %7.1f us   one at a time %7.1f us" % (B, N, name, timed(f), timed(f, sync=True)),
              flush=True)
