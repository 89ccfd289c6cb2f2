"""Randomised parity soak of the small-batch kernels (knn_small.hip, ball_small.hip, fps_small_kernel) and of the paths
they replaced, against the CPU oracle: random batch sizes, ragged lengths (zeros included), D, K, norms, lattices for
ties.  Every kernel choice a knob can force is run on every case and must reproduce the oracle bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _cloud(rng, shape, lattice):
    if lattice:
        return (rng.integers(0, 5, shape).astype(np.float32) * np.float32(0.25)).astype(np.float32)
    return rng.random(shape, dtype=np.float32)


def _lengths(rng, N, P):
    L = rng.integers(0, P + 1, N)
    L[rng.integers(0, N)] = P
    if rng.random() < 0.2:
        L[rng.integers(0, N)] = 0
    return L.astype(np.int64)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_knn_small(dev, oracle, monkeypatch, seed):
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(100 + seed)
    for it in range(25):
        lattice = rng.random() < 0.3
        N, P1, P2 = int(rng.integers(1, 6)), int(rng.integers(1, 700)), int(rng.integers(1, 3000))
        D, K = int(rng.integers(1, 9)), int(rng.choice([1, 2, 3, 4, 7, 8, 15, 16, 24, 32]))
        norm = int(rng.integers(1, 3))
        p1, p2 = _cloud(rng, (N, P1, D), lattice), _cloud(rng, (N, P2, D), lattice)
        l1, l2 = _lengths(rng, N, P1), _lengths(rng, N, P2)
        oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
        for knob, version in (("knn_small=1,knn_small_q=1", 2), ("knn_small=1,knn_small_q=2", 2), ("knn_small=0", 2),
                              ("", -1)):
            monkeypatch.setenv("POINTOPS_DEBUG", knob)
            i, d = _C.knn_points_idx(_G(p1, dev), _G(p2, dev), _G(l1, dev), _G(l2, dev), norm, K, version)
            what = dict(N=N, P1=P1, P2=P2, D=D, K=K, norm=norm, lattice=lattice, knob=knob, l1=l1.tolist(), l2=l2.tolist())
            assert np.array_equal(i.cpu().numpy(), oi), what
            assert np.array_equal(_bits(d.cpu().numpy()), _bits(od)), what


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_ball_small(dev, oracle, monkeypatch, seed):
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(200 + seed)
    for it in range(25):
        lattice = rng.random() < 0.3
        N, P1, P2 = int(rng.integers(1, 6)), int(rng.integers(1, 700)), int(rng.integers(1, 3000))
        D, K = int(rng.integers(1, 7)), int(rng.choice([1, 3, 8, 16, 32, 64, 100, 500]))
        radius = float(rng.choice([0.05, 0.2, 0.5, 0.26, 1.5]))
        p1, p2 = _cloud(rng, (N, P1, D), lattice), _cloud(rng, (N, P2, D), lattice)
        l1, l2 = _lengths(rng, N, P1), _lengths(rng, N, P2)
        oi, od = oracle.ball_query(p1, p2, l1, l2, K, radius)
        for knob in ("ball_small=1", "ball_small=0"):
            monkeypatch.setenv("POINTOPS_DEBUG", knob)
            i, d = _C.ball_query(_G(p1, dev), _G(p2, dev), _G(l1, dev), _G(l2, dev), K, radius)
            what = dict(N=N, P1=P1, P2=P2, D=D, K=K, r=radius, lattice=lattice, knob=knob, l1=l1.tolist(), l2=l2.tolist())
            assert np.array_equal(i.cpu().numpy(), oi), what
            assert np.array_equal(_bits(d.cpu().numpy()), _bits(od)), what


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_fps_small(dev, oracle, monkeypatch, seed):
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(300 + seed)
    for it in range(25):
        lattice = rng.random() < 0.3
        N, P, D = int(rng.integers(1, 6)), int(rng.integers(1, 5000)), int(rng.choice([2, 3]))
        pts = _cloud(rng, (N, P, D), lattice)
        L = _lengths(rng, N, P)
        Kt = rng.integers(0, min(P, 200) + 1, N).astype(np.int64)
        S = np.array([int(rng.integers(0, max(int(v), 1))) for v in L], dtype=np.int64)
        want = oracle.sample_farthest_points(pts, L, Kt, S)
        for knob in ("", "fps_small=0"):
            monkeypatch.setenv("POINTOPS_DEBUG", knob)
            got = _C.sample_farthest_points(_G(pts, dev), _G(L, dev), _G(Kt, dev), _G(S, dev)).cpu().numpy()
            what = dict(N=N, P=P, D=D, lattice=lattice, knob=knob, L=L.tolist(), K=Kt.tolist(), S=S.tolist())
            assert np.array_equal(got, want), what


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_knn_grid(dev, oracle, monkeypatch, seed):
    """The grid family (version 3) on random mid-size shapes: ragged lengths with empty clouds, D = 1..3, every list
    size class, lattices (ties), self-queries; the quad pass forced on and off."""
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(400 + seed)
    for it in range(10):
        lattice = rng.random() < 0.3
        N, P1, P2 = int(rng.integers(1, 5)), int(rng.integers(1, 3000)), int(rng.integers(1, 6000))
        D, K = int(rng.integers(1, 4)), int(rng.choice([1, 2, 4, 8, 16, 20, 32, 40, 64, 100]))
        norm = int(rng.integers(1, 3))
        same = rng.random() < 0.25
        p2 = _cloud(rng, (N, P2, D), lattice)
        l2 = _lengths(rng, N, P2)
        p1, l1 = (p2, l2) if same else (_cloud(rng, (N, P1, D), lattice), _lengths(rng, N, P1))
        oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
        t2, tl2 = _G(p2, dev), _G(l2, dev)
        t1, tl1 = (t2, tl2) if same else (_G(p1, dev), _G(l1, dev))
        for knob in ("grid_quad=1", "grid_quad=0"):
            monkeypatch.setenv("POINTOPS_DEBUG", knob)
            i, d = _C.knn_points_idx(t1, t2, tl1, tl2, norm, K, 3)
            what = dict(N=N, P1=p1.shape[1], P2=P2, D=D, K=K, norm=norm, lattice=lattice, same=same, knob=knob,
                        l1=l1.tolist(), l2=l2.tolist())
            assert np.array_equal(i.cpu().numpy(), oi), what
            assert np.array_equal(_bits(d.cpu().numpy()), _bits(od)), what


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_ball_grid_and_fps_clusters(dev, oracle, monkeypatch, seed):
    """Ball query through the cell grid (forced) and FPS through multi-workgroup clusters on random ragged batches."""
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(500 + seed)
    for it in range(8):
        lattice = rng.random() < 0.25
        N, P1, P2 = int(rng.integers(1, 5)), int(rng.integers(1, 3000)), int(rng.integers(1, 6000))
        D, K = int(rng.integers(1, 4)), int(rng.choice([1, 4, 16, 32, 64]))
        radius = float(rng.choice([0.02, 0.05, 0.1, 0.3]))
        p1, p2 = _cloud(rng, (N, P1, D), lattice), _cloud(rng, (N, P2, D), lattice)
        l1, l2 = _lengths(rng, N, P1), _lengths(rng, N, P2)
        oi, od = oracle.ball_query(p1, p2, l1, l2, K, radius)
        monkeypatch.setenv("POINTOPS_DEBUG", "ball_grid=1,ball_factor=0")
        i, d = _C.ball_query(_G(p1, dev), _G(p2, dev), _G(l1, dev), _G(l2, dev), K, radius)
        what = dict(N=N, P1=P1, P2=P2, D=D, K=K, r=radius, lattice=lattice, l1=l1.tolist(), l2=l2.tolist())
        assert np.array_equal(i.cpu().numpy(), oi), what
        assert np.array_equal(_bits(d.cpu().numpy()), _bits(od)), what
    monkeypatch.delenv("POINTOPS_DEBUG")
    for it in range(6):
        N, P, D = int(rng.integers(1, 5)), int(rng.integers(4097, 30000)), int(rng.choice([2, 3]))
        pts = _cloud(rng, (N, P, D), rng.random() < 0.25)
        L = _lengths(rng, N, P)
        Kt = rng.integers(0, 80, N).astype(np.int64)
        S = np.array([int(rng.integers(0, max(int(v), 1))) for v in L], dtype=np.int64)
        want = oracle.sample_farthest_points(pts, L, Kt, S)
        got = _C.sample_farthest_points(_G(pts, dev), _G(L, dev), _G(Kt, dev), _G(S, dev)).cpu().numpy()
        assert np.array_equal(got, want), dict(N=N, P=P, D=D, L=L.tolist(), K=Kt.tolist(), S=S.tolist())
