#!/usr/bin/env python3
"""Host-side cost of one small knn_points call (BASELINE.json configs[0]: B=2, N=M=1024, K=8)."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import synth
from pytorch3d_pointops_amd.functions import knn_points
dev = torch.device("cuda:0")
a = torch.from_numpy(synth.uniform_f32(1, (2, 1024, 3))).to(dev)
for _ in range(20):
    knn_points(a, a, K=8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    knn_points(a, a, K=8)
torch.cuda.synchronize()
print("us per call (async loop):", (time.perf_counter() - t0) / 2000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000):
    knn_points(a, a, K=8)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
