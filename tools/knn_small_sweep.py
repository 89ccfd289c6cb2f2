"""knn_small.hip (one wave per query) against the sliced lane-per-query scan (knn.hip) and the grid, over small and
mid-size batches: the data behind knn_small_applies().  Times are back-to-back averages (10 calls per event pair)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=5, inner=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    return sorted(ts)[len(ts) // 2]


shapes = [(2, 1024, 1024, 3), (1, 256, 256, 3), (8, 512, 512, 3), (1, 4096, 4096, 3), (4, 4096, 4096, 3), (2, 8192, 8192, 3),
          (1, 16384, 16384, 3), (8, 2048, 2048, 3), (32, 1024, 1024, 3), (64, 2048, 2048, 3), (1, 1024, 65536, 3),
          (1, 512, 262144, 3), (16, 1024, 1024, 8), (1, 2048, 16384, 6)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
for (B, P1, P2, D) in shapes:
    x = torch.from_numpy(synth.uniform_f32(5, (B, P1, D))).to(dev)
    y = torch.from_numpy(synth.uniform_f32(6, (B, P2, D))).to(dev)
    L1 = torch.full((B,), P1, dtype=torch.int64, device=dev)
    L2 = torch.full((B,), P2, dtype=torch.int64, device=dev)
    for K in (1, 8, 32):
        r = {}
        for name, knob, v in (("wave_1", "knn_small=1,knn_small_q=1", 2), ("wave_q", "knn_small=1,knn_small_q=2", 2),
                              ("lane_per_query", "knn_small=0", 2), ("auto", "", -1)):
            os.environ["POINTOPS_DEBUG"] = knob
            r[name] = round(timeit(lambda: _C.knn_points_idx(x, y, L1, L2, 2, K, v)) * 1e3, 1)
        if D <= 3 and P2 >= 4096:
            os.environ["POINTOPS_DEBUG"] = ""
            r["grid"] = round(timeit(lambda: _C.knn_points_idx(x, y, L1, L2, 2, K, 3)) * 1e3, 1)
        print(json.dumps({"B": B, "P1": P1, "P2": P2, "D": D, "K": K, "us": r}), flush=True)
