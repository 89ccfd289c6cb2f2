#!/usr/bin/env python3
"""cProfile of the Python side of one small call (B=2, N=1024): python tools/host_profile.py ball|chamfer|chamfer_bwd|knn_bwd"""
import cProfile, io, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import synth
from pytorch3d_pointops_amd.functions import ball_query, knn_points
from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance
dev = torch.device("cuda:0")
a = torch.from_numpy(synth.uniform_f32(1, (2, 1024, 3))).to(dev)
b = torch.from_numpy(synth.uniform_f32(2, (2, 1024, 3))).to(dev)
ag, bg = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
def chamfer_bwd():
    loss, _ = chamfer_distance(ag, bg); loss.backward()
def knn_bwd():
    knn_points(ag, bg, K=8).dists.sum().backward()
fns = {"ball": lambda: ball_query(a, b, K=32, radius=0.2), "chamfer": lambda: chamfer_distance(a, b),
       "chamfer_bwd": chamfer_bwd, "knn_bwd": knn_bwd}
for which in sys.argv[1:]:
    f = fns[which]
    for _ in range(50): f()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(1000): f()
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("=====", which); print(s.getvalue()[:5500])
