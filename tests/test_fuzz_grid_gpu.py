"""Randomised parity soak of the GRID family of knn_points (cell grid, lane / radius-2 / box / wave searches, refined
cells, long lists) at sizes the CPU oracle does not finish in seconds: version 3 with the radius-2 pass forced on and off
against the brute-force families of the same library (versions 0 / 2, themselves pinned by the oracle and the reference's
goldens in test_gpu_parity.py and test_fuzz_small_gpu.py), bit for bit, on random distributions -- uniform, clusters of
very different density, lattices (ties), thin slabs, u^4 -- with ragged lengths.
POINTOPS_FUZZ_SOAK=<int> shifts the seeds for soak runs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_SOAK = int(os.environ.get("POINTOPS_FUZZ_SOAK", "0"))


def _draw(rng, n, p, d, kind):
    if kind == "uniform":
        x = rng.random((n, p, d), dtype=np.float32)
    elif kind == "clusters":  # a few gaussian blobs whose widths differ by orders of magnitude, plus background
        x = rng.random((n, p, d), dtype=np.float32)
        for b in range(n):
            start = 0
            for _ in range(int(rng.integers(1, 5))):
                m = int(rng.integers(1, max(2, p // 2)))
                c, w = rng.random(d), 10.0 ** rng.uniform(-4, -1)
                seg = slice(start, min(p, start + m))
                x[b, seg] = (c + w * rng.standard_normal((seg.stop - seg.start, d))).astype(np.float32)
                start = seg.stop
                if start >= p:
                    break
            x[b] = x[b][rng.permutation(p)]
    elif kind == "lattice":
        x = (rng.integers(0, int(rng.integers(3, 40)), (n, p, d)).astype(np.float32) * np.float32(0.125))
    elif kind == "slab":
        x = rng.random((n, p, d), dtype=np.float32)
        x[..., -1] *= np.float32(10.0 ** rng.uniform(-4, -1))
    else:  # u4
        x = rng.random((n, p, d), dtype=np.float32) ** 4
    return np.ascontiguousarray(x, dtype=np.float32)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_grid_family_vs_brute_force(dev, monkeypatch, seed):
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(900 + seed + _SOAK)
    kinds = ["uniform", "clusters", "lattice", "slab", "u4"]
    for it in range(8):
        d = int(rng.choice([1, 2, 3, 3, 3]))
        n = int(rng.integers(1, 5))
        p1, p2 = int(rng.integers(500, 40000)), int(rng.integers(500, 60000))
        K = int(rng.choice([1, 3, 8, 16, 16, 32, 40, 64, 100]))
        norm = int(rng.integers(1, 3))
        k1, k2 = kinds[int(rng.integers(0, 5))], kinds[int(rng.integers(0, 5))]
        a, b = _draw(rng, n, p1, d, k1), _draw(rng, n, p2, d, k2)
        if rng.random() < 0.3:
            b = b * np.float32(0.5) + np.float32(2.0)  # disjoint clouds: every query is far from every point
        l1 = rng.integers(1, p1 + 1, n).astype(np.int64)
        l2 = rng.integers(1, p2 + 1, n).astype(np.int64)
        l1[int(rng.integers(0, n))] = p1
        l2[int(rng.integers(0, n))] = p2
        if rng.random() < 0.3:
            l2[int(rng.integers(0, n))] = int(rng.integers(1, K + 2))  # fewer points than K (zero padding)
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        t1, t2 = torch.from_numpy(l1).to(dev), torch.from_numpy(l2).to(dev)
        monkeypatch.delenv("POINTOPS_DEBUG", raising=False)
        want_i, want_d = _C.knn_points_idx(ta, tb, t1, t2, norm, K, 0)
        what = dict(seed=seed, it=it, d=d, n=n, p1=p1, p2=p2, K=K, norm=norm, k1=k1, k2=k2, l1=l1.tolist(), l2=l2.tolist())
        for knob in ("grid_quad=1", "grid_quad=0", ""):
            if knob:
                monkeypatch.setenv("POINTOPS_DEBUG", knob)
            else:
                monkeypatch.delenv("POINTOPS_DEBUG", raising=False)
            for version in ((3,) if knob else (-1, 2 if K <= 32 else 0)):
                if not _C.knn_check_version(version, d, K) and version not in (-1,):
                    continue
                got_i, got_d = _C.knn_points_idx(ta, tb, t1, t2, norm, K, version)
                assert torch.equal(got_i, want_i), dict(what, knob=knob, version=version)
                assert torch.equal(got_d.view(torch.int32), want_d.view(torch.int32)), dict(what, knob=knob, version=version)
        # self-query through the same tensors (one sort)
        monkeypatch.delenv("POINTOPS_DEBUG", raising=False)
        si, sd = _C.knn_points_idx(tb, tb, t2, t2, norm, K, 3) if _C.knn_check_version(3, d, K) else (None, None)
        if si is not None:
            wi, wd = _C.knn_points_idx(tb, tb.clone(), t2, t2.clone(), norm, K, 0)
            assert torch.equal(si, wi) and torch.equal(sd.view(torch.int32), wd.view(torch.int32)), dict(what, self_query=True)


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_ball_query_grid_vs_scan(dev, monkeypatch, seed):
    """ball_query: the cell-grid path and the coarse-cell query order (forced wherever the shape allows, and by the
    automatic rule) against the storage-order scan without workspace (`ball_grid=0`, pinned by the oracle and the
    reference's goldens elsewhere), bit for bit, radii from a fraction of the point spacing to the whole cloud."""
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(950 + seed + _SOAK)
    kinds = ["uniform", "clusters", "lattice", "slab", "u4"]
    for it in range(8):
        d = int(rng.choice([1, 2, 3, 3, 3]))
        n = int(rng.integers(1, 5))
        p1, p2 = int(rng.integers(300, 30000)), int(rng.integers(300, 50000))
        K = int(rng.choice([1, 4, 8, 16, 32, 33, 64]))
        radius = float(10.0 ** rng.uniform(-3, 0.3))
        k1, k2 = kinds[int(rng.integers(0, 5))], kinds[int(rng.integers(0, 5))]
        a, b = _draw(rng, n, p1, d, k1), _draw(rng, n, p2, d, k2)
        l1 = rng.integers(1, p1 + 1, n).astype(np.int64)
        l2 = rng.integers(1, p2 + 1, n).astype(np.int64)
        l1[int(rng.integers(0, n))] = p1
        l2[int(rng.integers(0, n))] = p2
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        t1, t2 = torch.from_numpy(l1).to(dev), torch.from_numpy(l2).to(dev)
        monkeypatch.setenv("POINTOPS_DEBUG", "ball_grid=0")
        want_i, want_d = _C.ball_query(ta, tb, t1, t2, K, radius)
        what = dict(seed=seed, it=it, d=d, n=n, p1=p1, p2=p2, K=K, radius=radius, k1=k1, k2=k2, l1=l1.tolist(), l2=l2.tolist())
        for knob in ("ball_grid=1", "ball_grid=1,ball_order=0", "ball_grid=1,ball_stage=0", ""):
            if knob:
                monkeypatch.setenv("POINTOPS_DEBUG", knob)
            else:
                monkeypatch.delenv("POINTOPS_DEBUG", raising=False)
            got_i, got_d = _C.ball_query(ta, tb, t1, t2, K, radius)
            assert torch.equal(got_i, want_i), dict(what, knob=knob)
            assert torch.equal(got_d.view(torch.int32), want_d.view(torch.int32)), dict(what, knob=knob)
