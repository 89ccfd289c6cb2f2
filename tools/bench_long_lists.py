import sys, os, json, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from pytorch3d_pointops_amd import _C, synth
from bench_ops import timeit
dev = torch.device("cuda:0")
B, N = 32, 65536
x = torch.from_numpy(synth.uniform_f32(71, (B, N, 3))).to(dev)
y = torch.from_numpy(synth.uniform_f32(72, (B, N, 3))).to(dev)
L = torch.full((B,), N, dtype=torch.int64, device=dev)
for K in (32, 40, 64):
    ms, mn = timeit(lambda: _C.knn_points_idx(x, y, L, L, 2, K, -1), warmup=2, iters=5)
    print(json.dumps(dict(op=f"knn_points B=32 N=M=65536 K={K} (auto)", median_ms=ms, min_ms=mn)), flush=True)
x4, y4, L4 = x[:4, :16384].contiguous(), y[:4, :16384].contiguous(), torch.full((4,), 16384, dtype=torch.int64, device=dev)
for v in (-1, 0):
    ms, mn = timeit(lambda: _C.knn_points_idx(x4, y4, L4, L4, 2, 64, v), warmup=1, iters=3)
    print(json.dumps(dict(op=f"knn_points B=4 N=M=16384 K=64 version={v}", median_ms=ms, min_ms=mn)), flush=True)
