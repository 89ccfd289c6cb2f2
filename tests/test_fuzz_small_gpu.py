"""Randomised parity soak of the small-batch kernels (knn_small.hip, ball_small.hip, fps_small_kernel) and of the paths
they replaced, against the CPU oracle: random batch sizes, ragged lengths (zeros included), D, K, norms, lattices for
ties.  Every kernel choice a knob can force is run on every case and must reproduce the oracle bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# POINTOPS_FUZZ_SOAK=<int> shifts every seed: `for s in 1000 2000 ...; do POINTOPS_FUZZ_SOAK=$s pytest tests/test_fuzz_small_gpu.py; done`
# draws fresh cases (the committed run, offset 0, is what the suite pins).
import os  # noqa: E402

_SOAK = int(os.environ.get("POINTOPS_FUZZ_SOAK", "0"))


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

    rng = np.random.default_rng(100 + seed + _SOAK)
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

    rng = np.random.default_rng(200 + seed + _SOAK)
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

    rng = np.random.default_rng(300 + seed + _SOAK)
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

    rng = np.random.default_rng(400 + seed + _SOAK)
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

    rng = np.random.default_rng(500 + seed + _SOAK)
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


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_backward_and_gather(dev, oracle, monkeypatch, seed):
    """knn_points_backward (the KNN table of a real search, and a ball-query table with -1 padding) in its three
    forms -- LDS tiles / device atomics by shape, forced atomics, the deterministic inverted table -- against the oracle:
    grad_p1 bit-exact (register sums in k order), grad_p2 within 1e-5 (bit-exact in the deterministic form);
    knn_gather / masked_gather forward against their numpy restatements."""
    from oracle import oracle as O
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(600 + seed + _SOAK)
    for it in range(12):
        N, P1, P2 = int(rng.integers(1, 5)), int(rng.integers(1, 900)), int(rng.integers(1, 2500))
        D, K, norm = int(rng.integers(1, 7)), int(rng.choice([1, 3, 8, 16, 33])), int(rng.integers(1, 3))
        lattice = rng.random() < 0.3
        p1, p2 = _cloud(rng, (N, P1, D), lattice), _cloud(rng, (N, P2, D), lattice)
        l1, l2 = _lengths(rng, N, P1), _lengths(rng, N, P2)
        if it % 2 == 0:
            idx, _ = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
        else:
            idx, _ = oracle.ball_query(p1, p2, l1, l2, K, 0.3)
            norm = 2
        g = rng.standard_normal((N, P1, K)).astype(np.float32)
        g[rng.random((N, P1, K)) < 0.2] = 0.0
        w1, w2 = oracle.knn_points_backward(p1, p2, l1, l2, idx, norm, g)
        for knob, det in (("", False), ("knn_bwd_mode=a", False), ("knn_bwd_mode=t", False), ("", True)):
            monkeypatch.setenv("POINTOPS_DEBUG", knob)
            g1, g2 = _C.knn_points_backward(_G(p1, dev), _G(p2, dev), _G(l1, dev), _G(l2, dev), _G(idx, dev), norm,
                                            _G(g, dev), deterministic=det)
            what = dict(N=N, P1=P1, P2=P2, D=D, K=K, norm=norm, knob=knob, det=det, it=it)
            assert np.array_equal(_bits(g1.cpu().numpy()), _bits(w1)), what
            if det:
                assert np.array_equal(_bits(g2.cpu().numpy()), _bits(w2)), what
            else:
                # (atomically accumulated: the order differs from the oracle's; the deterministic form above pins the values)
                assert np.allclose(g2.cpu().numpy(), w2, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(w2).max()))), what
        monkeypatch.delenv("POINTOPS_DEBUG")
        U = int(rng.integers(1, 9))
        x = _cloud(rng, (N, P2, U), False)
        if it % 2 == 0:
            out = _C.gather_neighbors(_G(x, dev), _G(idx, dev), _G(l2, dev))
            assert np.array_equal(out.cpu().numpy(), O.knn_gather(x, idx, l2)), dict(it=it, U=U)
        else:
            out = _C.gather_neighbors(_G(x, dev), _G(idx, dev), None)
            assert np.array_equal(out.cpu().numpy(), O.masked_gather(x, idx)), dict(it=it, U=U)


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_wide_families_and_packing(dev, oracle, seed):
    """Feature-space D and long lists off the grid (families 0 / 2 by shape), the K > 64 grid search, packed <-> padded
    and sample_pdf on random ragged inputs, against the oracle."""
    from pytorch3d_pointops_amd import _C

    rng = np.random.default_rng(700 + seed + _SOAK)
    for it in range(10):
        N, P1, P2 = int(rng.integers(1, 4)), int(rng.integers(1, 400)), int(rng.integers(1, 1500))
        D, K = int(rng.choice([1, 3, 4, 9, 16, 33, 70])), int(rng.choice([1, 5, 32, 33, 64, 65, 100, 130]))
        norm = int(rng.integers(1, 3))
        lattice = rng.random() < 0.3
        p1, p2 = _cloud(rng, (N, P1, D), lattice), _cloud(rng, (N, P2, D), lattice)
        l1, l2 = _lengths(rng, N, P1), _lengths(rng, N, P2)
        oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
        versions = [-1, 0] + ([3] if D <= 3 and K <= 128 else [])
        for v in versions:
            i, d = _C.knn_points_idx(_G(p1, dev), _G(p2, dev), _G(l1, dev), _G(l2, dev), norm, K, v)
            what = dict(N=N, P1=P1, P2=P2, D=D, K=K, norm=norm, lattice=lattice, version=v, l1=l1.tolist(), l2=l2.tolist())
            assert np.array_equal(i.cpu().numpy(), oi), what
            assert np.array_equal(_bits(d.cpu().numpy()), _bits(od)), what
    for it in range(10):
        B, D = int(rng.integers(1, 7)), int(rng.integers(1, 6))
        sizes = rng.integers(0, 300, B)
        first = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        F = int(sizes.sum())
        if F == 0:
            continue
        M = int(max(sizes.max(), 1))
        packed = rng.standard_normal((F, D)).astype(np.float32)
        want = oracle.packed_to_padded(packed, first, M)
        got = _C.packed_to_padded(_G(packed, dev), _G(first, dev), M)
        assert np.array_equal(_bits(got.cpu().numpy()), _bits(want)), dict(B=B, D=D, sizes=sizes.tolist())
        back = _C.padded_to_packed(got, _G(first, dev), F)
        assert np.array_equal(_bits(back.cpu().numpy()), _bits(oracle.padded_to_packed(want, first, F)))
        assert np.array_equal(_bits(back.cpu().numpy()), _bits(packed))
    for it in range(8):
        batch, nb, ns = int(rng.integers(1, 40)), int(rng.integers(1, 70)), int(rng.integers(1, 90))
        bins = np.sort(rng.random((batch, nb + 1), dtype=np.float32), axis=1)
        w = rng.random((batch, nb), dtype=np.float32)
        w[rng.random((batch, nb)) < 0.2] = 0.0
        u = rng.random((batch, ns), dtype=np.float32)
        eps = float(rng.choice([1e-5, 1e-2, 0.0]))
        want = oracle.sample_pdf(bins, w, u.copy(), eps)
        out = _G(u.copy(), dev)
        _C.sample_pdf(_G(bins, dev), _G(w, dev), out, eps)
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(want)), dict(batch=batch, nb=nb, ns=ns, eps=eps)


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_chamfer_native_vs_composed(dev, monkeypatch, seed):
    """The one-call bidirectional chamfer against the composed path (knn_points + knn_gather + torch reductions) on
    random ragged batches, reductions, norms and optional normals: losses and all gradients."""
    import pytorch3d_pointops_amd.functions.chamfer as ch

    rng = np.random.default_rng(800 + seed + _SOAK)
    for it in range(8):
        N, P1, P2 = int(rng.integers(1, 6)), int(rng.integers(2, 900)), int(rng.integers(2, 1200))
        l1 = np.maximum(_lengths(rng, N, P1), 1)
        l2 = np.maximum(_lengths(rng, N, P2), 1)
        with_normals = rng.random() < 0.5
        br = [None, "mean", "sum"][int(rng.integers(0, 3))]
        pr = ["mean", "sum"][int(rng.integers(0, 2))]
        norm = int(rng.integers(1, 3))
        abs_cos = bool(rng.random() < 0.5)
        base = dict(x=_cloud(rng, (N, P1, 3), False), y=_cloud(rng, (N, P2, 3), False))
        if with_normals:
            base.update(xn=_cloud(rng, (N, P1, 3), False) - np.float32(0.5), yn=_cloud(rng, (N, P2, 3), False) - np.float32(0.5))

        def run():
            t = {k: _G(v, dev).requires_grad_(True) for k, v in base.items()}
            kw = dict(x_features={"normals": t["xn"]}, y_features={"normals": t["yn"]}, feature_names=["normals"]) \
                if with_normals else {}
            loss, lf = ch.chamfer_distance(t["x"], t["y"], x_lengths=_G(l1, dev), y_lengths=_G(l2, dev), batch_reduction=br,
                                           point_reduction=pr, norm=norm, abs_cosine=abs_cos, **kw)
            total = loss.sum() * 1.25 + (lf["normals"].sum() * 0.5 if with_normals else 0.0)
            total.backward()
            return [loss.detach()] + ([lf["normals"].detach()] if with_normals else []), {k: v.grad for k, v in t.items()}

        a = run()
        with monkeypatch.context() as m:
            m.setattr(ch, "_fused_direction_ok", lambda *args, **kw: False)
            b = run()
        what = dict(N=N, P1=P1, P2=P2, br=br, pr=pr, norm=norm, normals=with_normals, l1=l1.tolist(), l2=l2.tolist())
        for u, v in zip(a[0], b[0]):
            assert u.shape == v.shape, what
            assert np.allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-5, atol=1e-6), what
        for k in base:
            # (normals of length ~0 give cosine gradients of size ~1/|n| with cancellation inside, and a target normal
            # sums such terms from every point it is nearest to, in a different order on the two paths: the absolute
            # term scales with the largest gradient of the tensor)
            u, v = a[1][k].cpu().numpy(), b[1][k].cpu().numpy()
            # (and with a one-point cloud on one side that point's gradient is a sum of ~1000 atomically added terms)
            tol = 1e-4 * np.abs(v) + 1e-5 * max(1.0, float(np.abs(v).max()))
            assert (np.abs(u - v) <= tol).all(), (k, float((np.abs(u - v) - tol).max()), what)
