#!/usr/bin/env python3
"""chamfer fwd+bwd at a training-loop size (B=8, N=2048) a few hundred times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import synth
from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance
dev = torch.device("cuda:0")
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 2048)
a = torch.from_numpy(synth.uniform_f32(1, (B, N, 3))).to(dev).requires_grad_(True)
b = torch.from_numpy(synth.uniform_f32(2, (B, N, 3))).to(dev).requires_grad_(True)
for _ in range(20):
    chamfer_distance(a, b)[0].backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    chamfer_distance(a, b)[0].backward()
torch.cuda.synchronize()
print("us per fwd+bwd:", (time.perf_counter() - t0) / 300 * 1e6)
