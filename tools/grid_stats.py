#!/usr/bin/env python3
"""Fallback fraction of the grid KNN on uniform clouds for several K (diagnostics)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth
dev = torch.device("cuda:0")
for (B, P, K) in ((4, 65536, 16), (4, 65536, 1), (4, 65536, 8), (4, 65536, 4), (4, 65536, 32), (2, 200000, 1), (4, 16384, 16)):
    p1 = torch.from_numpy(synth.uniform_f32(1, (B, P, 3))).to(dev)
    p2 = torch.from_numpy(synth.uniform_f32(2, (B, P, 3))).to(dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    i, d, c = _C.knn_grid_fallback_counts(p1, p2, L, L, 2, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _C.knn_points_idx(p1, p2, L, L, 2, K, 3); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"B={B} P={P} K={K}: expanding {c[0].cpu().tolist()} = {100.0*float(c[0].sum())/(B*P):.2f}%  whole-cloud {c[1].cpu().tolist()}  time {min(ts):.3f} ms", flush=True)
