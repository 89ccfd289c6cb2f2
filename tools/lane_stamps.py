#!/usr/bin/env python3
"""Phase times of the lane search's chunks (diagnostic build: tools/build_variant.py stamps knn_grid_d3.hip
-DPOINTOPS_LANE_STAMPS, run with POINTOPS_AMD_LIB=<variant>): mean s_memtime ticks between the stamps."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

dev = torch.device("cuda:0")
B, P, K = 32, 65536, int(os.environ.get("K", 16))
p1 = torch.from_numpy(synth.uniform_f32(11, (B, P, 3))).to(dev)
p2 = torch.from_numpy(synth.uniform_f32(12, (B, P, 3))).to(dev)
L = torch.full((B,), P, dtype=torch.int64, device=dev)
for _ in range(3):
    _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
torch.cuda.synchronize()
n = 32768
buf = np.zeros((1 << 16, 8), dtype=np.int64)
fn = _C._lib.pointops_debug_lane_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
s = buf[:n]
names = ["decode+query", "pieces(cstart)", "stage+rows", "sync", "walk", "write"]
d = np.diff(s[:, :7], axis=1)
print("ticks (100 MHz s_memtime => x10 ns): mean / median / p90 per phase")
for k, nm in enumerate(names):
    print(f"  {nm:16s} {d[:, k].mean():9.1f} {np.median(d[:, k]):9.1f} {np.percentile(d[:, k], 90):9.1f}")
tot = s[:, 6] - s[:, 0]
print(f"  {'chunk':16s} {tot.mean():9.1f} {np.median(tot):9.1f} {np.percentile(tot, 90):9.1f}")
print("kernel span ticks:", s[:, 6].max() - s[:, 0].min())
