#!/usr/bin/env python3
"""Timing build `-DPOINTOPS_EXPERIMENT_QUAD_STAMPS` (POINTOPS_AMD_LIB=...qstamps.so): per wave of the radius-2 pass the clock
cycles of its prologue / walk / epilogue, its pipeline stages and its records per lane, read back from the rows it marks."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

dev = torch.device("cuda:0")
B, N, K = 32, 65536, 16
a = torch.from_numpy(synth.uniform_f32(3, (B, N, 3))).to(dev)
b = torch.from_numpy(synth.uniform_f32(4, (B, N, 3))).to(dev)
L = torch.full((B,), N, dtype=torch.int64, device=dev)
for _ in range(3):
    idx, d = _C.knn_points_idx(a, b, L, L, 2, K, -1)
torch.cuda.synchronize()
idx, d = idx.cpu().numpy().reshape(-1, K), d.cpu().numpy().reshape(-1, K).view(np.int32)
rows = np.nonzero(idx[:, 0] == -12345)[0]
s = d[rows]
print("marked waves:", len(rows))
for name, col in (("prologue cycles", 1), ("walk cycles", 2), ("epilogue cycles", 3), ("stages", 4), ("records lane 0", 5),
                  ("records, longest lane", 6)):
    v = s[:, col].astype(np.float64)
    print(f"{name:24s} mean {v.mean():10.1f}  median {np.median(v):10.1f}  p90 {np.percentile(v, 90):10.1f}  max {v.max():10.1f}")
t0 = s[:, 7].astype(np.int64)
t_end = t0 + s[:, 1] + s[:, 2] + s[:, 3]
print("span of wave starts (cycles):", int(t0.max() - t0.min()), " first start -> last end:", int(t_end.max() - t0.min()))
