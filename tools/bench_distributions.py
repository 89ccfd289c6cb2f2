#!/usr/bin/env python3
"""Grid KNN / chamfer on NON-UNIFORM clouds: cfg2-size knn_points K=16 and a cfg4-shape
chamfer fwd+bwd on a sphere surface, two noisy planes, u^4 clustering, half the cloud in a 1e-3 cube and a 100:1
slab, next to the uniform cube.  One JSON line per case: ms, ratio to uniform, cells per dimension of cloud 0,
queries per cloud (mean over clouds) that the lane pass / the quad pass could not certify and that ended in the
whole-cloud scan.

    python tools/bench_distributions.py [--clouds 32] [--points 65536] [--k 16]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402
from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance  # noqa: E402
from bench_ops import timeit  # noqa: E402


def clouds(name, seed, b, n, dev):
    return torch.from_numpy(np.stack([synth.distribution(name, seed + 10 * i, n) for i in range(b)])).to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=32)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--chamfer-clouds", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, P, K = args.clouds, args.points, args.k
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    base = {}
    for name in synth.DISTRIBUTIONS:
        p1, p2 = clouds(name, 9001, B, P, dev), clouds(name, 9002, B, P, dev)
        ms, mn = timeit(lambda: _C.knn_points_idx(p1, p2, L, L, 2, K, -1), warmup=2, iters=7)
        _, _, st = _C.knn_grid_stats(p1, p2, L, L, 2, K)
        st = st.cpu().numpy().astype(np.float64)
        base.setdefault("knn", ms if name == "uniform" else base.get("knn"))
        print(json.dumps(dict(op=f"knn_points B={B} N=M={P} K={K}", dist=name, median_ms=ms, min_ms=mn,
                              vs_uniform=ms / base["knn"], cells_cloud0=st[0, :3].astype(int).tolist(),
                              grid_used=int(st[:, 4].sum()), uncertified_lane=st[:, 5].mean(),
                              uncertified_quad_box=st[:, 6].mean(), whole_cloud_scan=st[:, 7].mean(),
                              deferred_to_box=st[:, 8].mean(), refined_cells=st[:, 9].mean())), flush=True)
        # self-query (p1 is p2): one sort
        ms_s, mn_s = timeit(lambda: _C.knn_points_idx(p2, p2, L, L, 2, K, -1), warmup=2, iters=7)
        print(json.dumps(dict(op=f"knn_points SELF B={B} N={P} K={K}", dist=name, median_ms=ms_s, min_ms=mn_s,
                              vs_p1_ne_p2=ms_s / ms)), flush=True)
    # cfg4 shape: ragged 20k..200k, normals, fwd + bwd
    Bc = args.chamfer_clouds
    l1 = synth.randint(41, 20000, 200000, (Bc,))
    l2 = synth.randint(42, 20000, 200000, (Bc,))
    P1, P2 = int(l1.max()), int(l2.max())
    for name in synth.DISTRIBUTIONS:
        x = clouds(name, 9101, Bc, P1, dev).requires_grad_(True)
        y = clouds(name, 9102, Bc, P2, dev).requires_grad_(True)
        xn = torch.from_numpy(synth.unit_normals(45, (Bc, P1, 3))).to(dev)
        yn = torch.from_numpy(synth.unit_normals(46, (Bc, P2, 3))).to(dev)
        xl, yl = torch.from_numpy(l1).to(dev), torch.from_numpy(l2).to(dev)

        def step():
            x.grad = y.grad = None
            loss, lf = chamfer_distance(x, y, x_lengths=xl, y_lengths=yl, x_features={"normals": xn},
                                        y_features={"normals": yn}, feature_names=["normals"])
            (loss + lf["normals"]).backward()

        ms, mn = timeit(step, warmup=2, iters=5)
        base.setdefault("chamfer", ms if name == "uniform" else base.get("chamfer"))
        print(json.dumps(dict(op=f"chamfer fwd+bwd B={Bc} ragged 20k..200k + normals", dist=name, median_ms=ms,
                              min_ms=mn, vs_uniform=ms / base["chamfer"])), flush=True)


if __name__ == "__main__":
    main()
