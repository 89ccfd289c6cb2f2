"""v2 (sliced all-pairs scan) against v3 (grid) over batch / cloud sizes / K: the data behind choose_version()."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for K in (1, 4, 8, 16, 32):
    for (B, N) in ((1, 4096), (4, 4096), (8, 4096), (16, 4096), (32, 4096), (1, 8192), (2, 8192), (4, 8192),
                   (16, 8192), (1, 16384), (2, 16384), (4, 16384), (1, 32768)):
        x = torch.from_numpy(synth.uniform_f32(5, (B, N, 3))).to(dev)
        y = torch.from_numpy(synth.uniform_f32(6, (B, N, 3))).to(dev)
        L = torch.full((B,), N, dtype=torch.int64, device=dev)
        r = {v: round(timeit(lambda: _C.knn_points_idx(x, y, L, L, 2, K, v)), 4) for v in (2, 3)}
        print(json.dumps({"K": K, "B": B, "N": N, "log2pairs": round(math.log2(B * N * N), 1), "v2": r[2],
                          "v3": r[3], "best": 2 if r[2] < r[3] else 3}), flush=True)
