#!/usr/bin/env python3
"""knn_points at the cfg2 cloud size (B=32, N=M=65536, D=3) for a range of K: one JSON line per K."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, P = 32, 65536
    p1 = torch.from_numpy(synth.uniform_f32(11, (B, P, 3))).to(dev)
    p2 = torch.from_numpy(synth.uniform_f32(12, (B, P, 3))).to(dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    for K in (1, 2, 4, 8, 16, 32):
        for _ in range(3):
            _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
        times = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        times.sort()
        print(json.dumps({"op": f"knn_points B={B} N=M={P} K={K}", "median_ms": times[len(times) // 2],
                          "min_ms": times[0]}), flush=True)


if __name__ == "__main__":
    main()
