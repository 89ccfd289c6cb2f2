#!/usr/bin/env python3
"""Secondary measurements at BASELINE.json configs[2..3] sizes (one GPU):
ball_query r=0.2 K=32 + sample_farthest_points K=1024 at B=16 N=131072, chamfer fwd+bwd at
B=8 ragged N in [20k,200k] with a "normals" feature, packed<->padded, knn_gather.
Prints one JSON line per op with HIP-event times and algorithmic GB/s.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import _C, synth  # noqa: E402
from pytorch3d_pointops_amd.functions import ball_query, knn_points, packed_to_padded, padded_to_packed  # noqa: E402
from pytorch3d_pointops_amd.functions import sample_farthest_points, knn_gather  # noqa: E402
from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance  # noqa: E402


def timeit(fn, warmup=2, iters=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def emit(name, ms, mn, **kw):
    print(json.dumps(dict(op=name, median_ms=ms, min_ms=mn, **kw)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="bq,fps,chamfer,packed,gather,cov,knn_bwd,knn_small")
    args = ap.parse_args()
    ops = args.ops.split(",")
    dev = torch.device("cuda:0")
    if "bq" in ops or "fps" in ops:
        Bc, Pc = 16, 131072
        pts = torch.from_numpy(synth.uniform_f32(31, (Bc, Pc, 3))).to(dev)
        L = torch.full((Bc,), Pc, dtype=torch.int64, device=dev)
    if "bq" in ops:
        ms, mn = timeit(lambda: _C.ball_query(pts, pts, L, L, 32, 0.2))
        outb = Bc * Pc * 32 * 12 + Bc * Pc * 12
        emit("ball_query B=16 N=131072 r=0.2 K=32", ms, mn, algo_GBs=outb / ms / 1e6)
        for (rr, kk, b) in ((0.05, 32, 16), (0.02, 32, 16), (0.01, 16, 16), (0.02, 32, 2)):
            for g in ("1", "0"):
                os.environ["POINTOPS_DEBUG"] = "ball_grid=" + g
                ms2, mn2 = timeit(lambda: _C.ball_query(pts[:b], pts[:b], L[:b], L[:b], kk, rr), warmup=1, iters=3)
                emit(f"ball_query B={b} N=131072 r={rr} K={kk} [grid={g}]", ms2, mn2)
        del os.environ["POINTOPS_DEBUG"]
        ms, mn = timeit(lambda: ball_query(pts, pts, L, L, K=32, radius=0.2, return_nn=True))
        emit("ball_query(+return_nn) B=16 N=131072", ms, mn, algo_GBs=(outb + Bc * Pc * 32 * 12) / ms / 1e6)
    if "fps" in ops:
        Kt = torch.full((Bc,), 1024, dtype=torch.int64, device=dev)
        S = torch.zeros((Bc,), dtype=torch.int64, device=dev)
        ms, mn = timeit(lambda: _C.sample_farthest_points(pts, L, Kt, S), warmup=1, iters=3)
        emit("sample_farthest_points B=16 N=131072 K=1024", ms, mn,
             point_updates_per_s=Bc * 1023 * Pc / ms * 1e3, streamed_GBs=Bc * 1023 * Pc * 20 / ms / 1e6)
        # cluster placement modes (debug.h): 0 round-robin members, 1 XCD-local members, 2 XCD-local + L2 exchange
        for mode in ("0", "1", "2"):
            os.environ["POINTOPS_DEBUG"] = "fps_mode=" + mode
            ms2, mn2 = timeit(lambda: _C.sample_farthest_points(pts, L, Kt, S), warmup=1, iters=5)
            emit(f"sample_farthest_points B=16 N=131072 K=1024 [fps_mode={mode}]", ms2, mn2, us_per_iteration=ms2 / 1.023)
        del os.environ["POINTOPS_DEBUG"]
        for (b, n, k) in ((1, 131072, 1024), (4, 32768, 512), (64, 8192, 256)):
            Kb = torch.full((b,), k, dtype=torch.int64, device=dev)
            Lb = torch.full((b,), n, dtype=torch.int64, device=dev)
            pb = pts[:b, :n].contiguous() if b <= Bc else pts[:, :n].repeat(b // Bc, 1, 1).contiguous()
            ms2, mn2 = timeit(lambda: _C.sample_farthest_points(pb, Lb, Kb, torch.zeros_like(Kb)), warmup=1, iters=3)
            emit(f"sample_farthest_points B={b} N={n} K={k}", ms2, mn2)
    if "chamfer" in ops:
        Bq = 8
        l1 = synth.randint(41, 20000, 200000, (Bq,))
        l2 = synth.randint(42, 20000, 200000, (Bq,))
        P1, P2 = int(l1.max()), int(l2.max())
        x = torch.from_numpy(synth.uniform_f32(43, (Bq, P1, 3))).to(dev).requires_grad_(True)
        y = torch.from_numpy(synth.uniform_f32(44, (Bq, P2, 3))).to(dev).requires_grad_(True)
        xn = torch.from_numpy(synth.unit_normals(45, (Bq, P1, 3))).to(dev).requires_grad_(True)
        yn = torch.from_numpy(synth.unit_normals(46, (Bq, P2, 3))).to(dev).requires_grad_(True)
        xl, yl = torch.from_numpy(l1).to(dev), torch.from_numpy(l2).to(dev)
        pairs = 2.0 * float((l1.astype(np.float64) * l2).sum())

        def fb():
            loss, lf = chamfer_distance(x, y, x_lengths=xl, y_lengths=yl, x_features={"normals": xn},
                                        y_features={"normals": yn}, feature_names=["normals"])
            (loss + lf["normals"]).backward()

        ms, mn = timeit(fb, warmup=1, iters=3)
        emit("chamfer fwd+bwd B=8 ragged [20k,200k] +normals", ms, mn, Mpairs_per_s=pairs / ms / 1e3,
             lengths1=l1.tolist(), lengths2=l2.tolist())
    if "packed" in ops:
        lens = synth.randint(51, 20000, 200000, (8,))
        F = int(lens.sum())
        first = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)).to(dev)
        xpk = torch.from_numpy(synth.uniform_f32(52, (F, 3))).to(dev)
        M = int(lens.max())
        ms, mn = timeit(lambda: packed_to_padded(xpk, first, M))
        emit(f"packed_to_padded F={F} D=3 max={M}", ms, mn, algo_GBs=(F * 12 + 8 * M * 12) / ms / 1e6)
        pad = packed_to_padded(xpk, first, M)
        ms, mn = timeit(lambda: padded_to_packed(pad, first, F))
        emit(f"padded_to_packed F={F} D=3", ms, mn, algo_GBs=(2 * F * 12) / ms / 1e6)
    if "gather" in ops:
        Bg, Pg, Kg = 32, 65536, 16
        xg = torch.from_numpy(synth.uniform_f32(61, (Bg, Pg, 3))).to(dev)
        ig = torch.from_numpy(synth.randint(62, 0, Pg - 1, (Bg, Pg, Kg))).to(dev)
        ms, mn = timeit(lambda: knn_gather(xg, ig))
        emit("knn_gather B=32 N=65536 K=16 U=3", ms, mn, algo_GBs=(Bg * Pg * Kg * (8 + 12) + Bg * Pg * 12) / ms / 1e6)
    if "gather" in ops:
        xw = torch.from_numpy(synth.uniform_f32(64, (8, 16384, 64))).to(dev)
        iw = torch.from_numpy(synth.randint(65, 0, 16383, (8, 16384, 16))).to(dev)
        ms, mn = timeit(lambda: knn_gather(xw, iw))
        emit("knn_gather B=8 N=16384 K=16 U=64", ms, mn, algo_GBs=(8 * 16384 * 16 * (8 + 256) + 8 * 16384 * 256) / ms / 1e6)
        gg = torch.from_numpy(synth.uniform_f32(63, (Bg, Pg, Kg, 3))).to(dev)
        for mode in ("tiled", "atomic"):
            os.environ["POINTOPS_DEBUG"] = "gather_bwd_mode=" + mode
            ms, mn = timeit(lambda: _C.gather_neighbors_backward(gg, ig, None, Pg))
            emit(f"knn_gather backward B=32 N=65536 K=16 U=3 [{mode}]", ms, mn,
                 algo_GBs=(Bg * Pg * Kg * (8 + 12) + Bg * Pg * 12) / ms / 1e6)
        del os.environ["POINTOPS_DEBUG"]
    if "cov" in ops:
        Bv, Pv, Kv = 8, 65536, 16
        kn = torch.from_numpy(synth.uniform_f32(91, (Bv, Pv, Kv, 3))).to(dev)

        def composed():
            m = kn.mean(2, keepdim=True)
            cd = kn - m
            return (cd.unsqueeze(4) * cd.unsqueeze(3)).mean(2)

        ms, mn = timeit(lambda: _C.point_covariances(kn))
        emit("point_covariances (fused) B=8 N=65536 K=16 D=3", ms, mn, algo_GBs=(Bv * Pv * (Kv * 12 + 36)) / ms / 1e6)
        ms, mn = timeit(composed)
        emit("point_covariances (composed torch ops, reference form)", ms, mn)
    if "knn_bwd" in ops:
        Bb, Pb, Kb = 32, 65536, 16
        a = torch.from_numpy(synth.uniform_f32(81, (Bb, Pb, 3))).to(dev)
        c = torch.from_numpy(synth.uniform_f32(82, (Bb, Pb, 3))).to(dev)
        Lb = torch.full((Bb,), Pb, dtype=torch.int64, device=dev)
        idx, _ = _C.knn_points_idx(a, c, Lb, Lb, 2, Kb, -1)
        gd = torch.from_numpy(synth.uniform_f32(83, (Bb, Pb, Kb))).to(dev)
        algo = Bb * Pb * Kb * (8 + 4) + 3 * Bb * Pb * 12 + Bb * Pb * 12  # idx + grad_dists, p1/p2 reads + 2 grads
        for mode in ("tiled", "atomic"):
            os.environ["POINTOPS_DEBUG"] = "knn_bwd_mode=" + mode
            ms, mn = timeit(lambda: _C.knn_points_backward(a, c, Lb, Lb, idx, 2, gd))
            emit(f"knn_points_backward B=32 N=65536 K=16 [{mode}]", ms, mn, algo_GBs=algo / ms / 1e6,
                 scatter_adds_per_s=Bb * Pb * Kb * 3 / ms * 1e3)
        del os.environ["POINTOPS_DEBUG"]
        ms, mn = timeit(lambda: _C.knn_points_backward(a, c, Lb, Lb, idx, 2, gd, deterministic=True), warmup=1, iters=5)
        emit("knn_points_backward B=32 N=65536 K=16 [deterministic: inverted table]", ms, mn, algo_GBs=algo / ms / 1e6)
        go = torch.from_numpy(synth.uniform_f32(84, (Bb, Pb, Kb, 3))).to(dev)
        ms, mn = timeit(lambda: _C.gather_neighbors_backward(go, idx, None, Pb, deterministic=True), warmup=1, iters=5)
        emit("knn_gather backward B=32 N=65536 K=16 U=3 [deterministic: inverted table]", ms, mn)
        for (b, n, k) in ((1, 65536, 16), (4, 16384, 16), (8, 65536, 1)):
            a2, c2 = a[:b, :n].contiguous(), c[:b, :n].contiguous()
            L2 = torch.full((b,), n, dtype=torch.int64, device=dev)
            idx2, _ = _C.knn_points_idx(a2, c2, L2, L2, 2, k, -1)
            gd2 = gd[:b, :n, :k].contiguous()
            for mode in ("tiled", "atomic"):
                os.environ["POINTOPS_DEBUG"] = "knn_bwd_mode=" + mode
                ms, mn = timeit(lambda: _C.knn_points_backward(a2, c2, L2, L2, idx2, 2, gd2))
                emit(f"knn_points_backward B={b} N={n} K={k} [{mode}]", ms, mn)
            del os.environ["POINTOPS_DEBUG"]
    if "knn_small" in ops:
        for (b, n, k) in ((2, 1024, 8), (32, 4096, 16), (8, 65536, 1)):
            a = torch.from_numpy(synth.uniform_f32(71, (b, n, 3))).to(dev)
            c = torch.from_numpy(synth.uniform_f32(72, (b, n, 3))).to(dev)
            ms, mn = timeit(lambda: knn_points(a, c, K=k))
            emit(f"knn_points B={b} N=M={n} K={k}", ms, mn, Mpairs_per_s=b * n * n / ms / 1e3)


if __name__ == "__main__":
    main()
