#!/usr/bin/env python3
"""Eager call vs HIP-graph replay (pytorch3d_pointops_amd/graphs.py), per call: back to back (host and GPU overlapped)
and one call at a time (synchronise after every call: the host's enqueue time is exposed).  One JSON line per case."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import graphs, synth  # noqa: E402
from pytorch3d_pointops_amd.functions import ball_query, knn_points, sample_farthest_points  # noqa: E402
from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance  # noqa: E402

dev = torch.device("cuda:0")


def measure(call, iters):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    single = []
    for _ in range(max(10, iters // 4)):
        torch.cuda.synchronize()
        a = time.perf_counter()
        call()
        torch.cuda.synchronize()
        single.append(time.perf_counter() - a)
    return dict(back_to_back_us=(t2 - t0) / iters * 1e6, host_enqueue_us=(t1 - t0) / iters * 1e6,
                one_at_a_time_us=float(np.median(single)) * 1e6)


def case(name, fn, inputs, backward, iters):
    if backward:
        wrt = [t for t in inputs if t.requires_grad]

        def eager():
            out = fn(*inputs)
            return torch.autograd.grad(out if isinstance(out, torch.Tensor) else out[0], wrt)
    else:
        def eager():
            with torch.no_grad():
                return fn(*inputs)
    e = measure(eager, iters)
    step = graphs.capture(fn, inputs, backward=backward)
    g = measure(step.replay, iters)
    print(json.dumps(dict(case=name, backward=backward, eager=e, graph=g,
                          one_at_a_time_speedup=e["one_at_a_time_us"] / g["one_at_a_time_us"],
                          back_to_back_speedup=e["back_to_back_us"] / g["back_to_back_us"])), flush=True)


def rnd(seed, shape, grad=False):
    return torch.from_numpy(synth.uniform_f32(seed, shape)).to(dev).requires_grad_(grad)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    # the reference's example size (BASELINE configs[0]) and the headline (configs[1])
    a, b = rnd(1, (2, 1024, 3)), rnd(2, (2, 1024, 3))
    case("knn_points B=2 N=1024 K=8 (cfg1)", lambda p, q: knn_points(p, q, K=8)[:2], (a, b), False, iters)
    a, b = rnd(3, (32, 65536, 3)), rnd(4, (32, 65536, 3))
    case("knn_points B=32 N=65536 K=16 (cfg2)", lambda p, q: knn_points(p, q, K=16)[:2], (a, b), False, max(iters // 4, 20))
    del a, b
    pts = rnd(5, (4, 4096, 3))

    def sample_group(p):
        centers, _ = sample_farthest_points(p, K=128)
        return ball_query(centers, p, K=32, radius=0.15, return_nn=False)[:2]

    case("FPS 4x4096->128 + ball_query K=32 (a set-abstraction layer's sampling + grouping)", sample_group, (pts,), False, iters)
    for B, n in ((4, 2048), (8, 4096)):
        x, y = rnd(6, (B, n, 3), True), rnd(7, (B, n, 3), True)
        case(f"chamfer fwd B={B} N={n}", lambda u, v: chamfer_distance(u, v)[0], (x, y), False, iters)
        case(f"chamfer fwd+bwd B={B} N={n}", lambda u, v: chamfer_distance(u, v)[0], (x, y), True, iters)
    # cfg4: ragged 20k..200k points with normals
    Bq = 8
    l1, l2 = synth.randint(41, 20000, 200000, (Bq,)), synth.randint(42, 20000, 200000, (Bq,))
    P1, P2 = int(l1.max()), int(l2.max())
    x, y = rnd(43, (Bq, P1, 3), True), rnd(44, (Bq, P2, 3), True)
    xn = torch.from_numpy(synth.unit_normals(45, (Bq, P1, 3))).to(dev).requires_grad_(True)
    yn = torch.from_numpy(synth.unit_normals(46, (Bq, P2, 3))).to(dev).requires_grad_(True)
    xl, yl = torch.from_numpy(l1).to(dev), torch.from_numpy(l2).to(dev)

    def cfg4(u, v, un, vn):
        loss, lf = chamfer_distance(u, v, x_lengths=xl, y_lengths=yl, x_features={"normals": un},
                                    y_features={"normals": vn}, feature_names=["normals"])
        return loss + lf["normals"]

    case("chamfer fwd+bwd B=8 ragged 20k..200k + normals (cfg4)", cfg4, (x, y, xn, yn), True, max(iters // 4, 20))


if __name__ == "__main__":
    main()
