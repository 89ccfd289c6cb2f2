"""Run the cfg2-shaped knn_points_backward a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

B, P, K = 32, 65536, 16
dev = torch.device("cuda:0")
a = torch.from_numpy(synth.uniform_f32(81, (B, P, 3))).to(dev)
c = torch.from_numpy(synth.uniform_f32(82, (B, P, 3))).to(dev)
L = torch.full((B,), P, dtype=torch.int64, device=dev)
idx, _ = _C.knn_points_idx(a, c, L, L, 2, K, -1)
gd = torch.from_numpy(synth.uniform_f32(83, (B, P, K))).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    _C.knn_points_backward(a, c, L, L, idx, 2, gd)
torch.cuda.synchronize()
