#!/usr/bin/env python3
"""Run one op of the config table a few times (for rocprofv3): python tools/run_op.py ball_query [iters]
   ops: knn_cfg1 | ball_query (cfg3: B=16 N=131072 r=0.2 K=32, self query) | fps (cfg3) | knn (cfg2)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

op = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
if op == "ball_query":
    B, P = 16, 131072
    pts = torch.from_numpy(synth.uniform_f32(3, (B, P, 3))).to(dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    fn = lambda: _C.ball_query(pts, pts, L, L, 32, 0.2)  # noqa: E731
elif op == "fps":
    B, P = 16, 131072
    pts = torch.from_numpy(synth.uniform_f32(3, (B, P, 3))).to(dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    Kt = torch.full((B,), 1024, dtype=torch.int64, device=dev)
    S = torch.zeros((B,), dtype=torch.int64, device=dev)
    fn = lambda: _C.sample_farthest_points(pts, L, Kt, S)  # noqa: E731
elif op in ("knn_reuse", "knn_reuse_new_queries"):  # cfg2 with the opt-in grid reuse: same target, same / alternating queries
    import pytorch3d_pointops_amd as po

    B, P = 32, 65536
    p1 = torch.from_numpy(synth.uniform_f32(11, (B, P, 3))).to(dev)
    p1b = p1.flip(0).contiguous()
    p2 = torch.from_numpy(synth.uniform_f32(12, (B, P, 3))).to(dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    po.set_grid_cache(True)
    qs = [p1, p1b] if op == "knn_reuse_new_queries" else [p1, p1]

    def fn():
        qs.reverse()
        return _C.knn_points_idx(qs[0], p2, L, L, 2, 16, -1)
elif op == "knn_cfg1":  # BASELINE.json configs[0]: B=2, N=M=1024, K=8, self query
    a = torch.from_numpy(synth.uniform_f32(1, (2, 1024, 3))).to(dev)
    L = torch.full((2,), 1024, dtype=torch.int64, device=dev)
    fn = lambda: _C.knn_points_idx(a, a, L, L, 2, 8, -1)  # noqa: E731
elif op == "knn_k100":
    B, P = 1, 300000
    p1 = torch.from_numpy(synth.uniform_f32(3811, (B, P, 3))).to(dev)
    p2 = torch.from_numpy(synth.uniform_f32(3812, (B, P, 3))).to(dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    fn = lambda: _C.knn_points_idx(p1, p2, L, L, 2, 100, -1)  # noqa: E731
else:
    B, P = 32, 65536
    p1 = torch.from_numpy(synth.uniform_f32(11, (B, P, 3))).to(dev)
    p2 = torch.from_numpy(synth.uniform_f32(12, (B, P, 3))).to(dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    fn = lambda: _C.knn_points_idx(p1, p2, L, L, 2, 16, -1)  # noqa: E731
for _ in range(2):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    fn()
torch.cuda.synchronize()
print(f"{op}: {(time.perf_counter() - t0) / iters * 1e3:.4f} ms per call", flush=True)
