"""Cell-occupancy sweep of the grid KNN per K (the debug knob grid_c_scale multiplies grid_tuning's target)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=7):
    fn()
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


B, N = 32, 65536
x = torch.from_numpy(synth.uniform_f32(71, (B, N, 3))).to(dev)
y = torch.from_numpy(synth.uniform_f32(72, (B, N, 3))).to(dev)
L = torch.full((B,), N, dtype=torch.int64, device=dev)
KS = tuple(int(k) for k in os.environ.get("POINTOPS_SWEEP_KS", "1,2,4,8,16,24,32").split(","))
SCALES = tuple(os.environ.get("POINTOPS_SWEEP_SCALES", "0.5,0.7,1.0,1.4,2.0,2.8,4.0").split(","))
for K in KS:
    out = {}
    for sc in SCALES:
        os.environ["POINTOPS_DEBUG"] = "grid_c_scale=" + sc
        out[sc] = round(timeit(lambda: _C.knn_points_idx(x, y, L, L, 2, K, 3)), 4)
    print(json.dumps({"K": K, "B": B, "N": N, "ms_by_scale": out}), flush=True)
