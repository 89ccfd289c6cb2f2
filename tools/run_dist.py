#!/usr/bin/env python3
"""Run knn_points_idx a few times on one named distribution (for rocprofv3 --kernel-trace --stats):
    python tools/run_dist.py half_in_cluster [clouds] [points] [K] [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
K = int(sys.argv[4]) if len(sys.argv) > 4 else 16
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0")
p1 = torch.from_numpy(np.stack([synth.distribution(name, 9001 + 10 * i, P) for i in range(B)])).to(dev)
p2 = torch.from_numpy(np.stack([synth.distribution(name, 9002 + 10 * i, P) for i in range(B)])).to(dev)
L = torch.full((B,), P, dtype=torch.int64, device=dev)
for _ in range(iters):
    _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
torch.cuda.synchronize()
print("done")
