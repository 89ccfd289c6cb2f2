"""HIP-graph capture of the hot path (pytorch3d_pointops_amd/graphs.py): a replay -- on the captured inputs and on new
data copied into them -- must give what the eager call gives (bit-exact where the eager op is deterministic, 1e-5
where it accumulates with atomics), for every operator family, for forward + backward of the chamfer loss, and with
the opt-in grid reuse switched on (a capture must bypass it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(seed, shape, dev):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g).to(dev)


def _same(a, b):
    return a.shape == b.shape and bool((a == b).all())


@pytest.mark.parametrize("N,P1,P2,K", [(2, 1024, 1024, 8), (3, 20000, 30000, 16), (1, 300, 5000, 40)])
def test_graph_knn_forward(dev, N, P1, P2, K):
    from pytorch3d_pointops_amd import graphs
    from pytorch3d_pointops_amd.functions import knn_points

    a, b = _cloud(1, (N, P1, 3), dev), _cloud(2, (N, P2, 3), dev)
    l1 = torch.full((N,), P1, dtype=torch.int64, device=dev)
    l2 = torch.full((N,), P2, dtype=torch.int64, device=dev)
    l2[0] = P2 // 2

    def fn(p, q, u, v):
        out = knn_points(p, q, lengths1=u, lengths2=v, K=K)
        return out.dists, out.idx

    step = graphs.capture(fn, (a, b, l1, l2))
    want = fn(a, b, l1, l2)
    got = step()
    assert _same(got[0], want[0]) and _same(got[1], want[1])
    # new points AND new lengths through the static buffers: every data-dependent decision is taken on the device
    a2, b2 = _cloud(3, (N, P1, 3), dev) * 0.5, _cloud(4, (N, P2, 3), dev) ** 3
    l2b = l2.clone()
    l2b[-1] = max(K // 2, 1)  # fewer points than K in the last cloud: zero padding
    want = fn(a2, b2, l1, l2b)
    got = step(a2, b2, l1, l2b)
    assert _same(got[0], want[0]) and _same(got[1], want[1])


def test_graph_ball_fps_gather(dev):
    from pytorch3d_pointops_amd import graphs
    from pytorch3d_pointops_amd.functions import ball_query, knn_gather, sample_farthest_points

    pts = _cloud(5, (4, 4096, 3), dev)

    def fn(p):
        centers, cidx = sample_farthest_points(p, K=128)  # (an int K: no device-to-host read)
        ball = ball_query(centers, p, K=32, radius=0.15, return_nn=False)
        feats = knn_gather(p, ball.idx.clamp(min=0))
        return centers, cidx, ball.dists, ball.idx, feats

    step = graphs.capture(fn, (pts,))
    for t in (pts.clone(), _cloud(6, (4, 4096, 3), dev) ** 2):
        want = fn(t)
        got = step(t)
        for g, w in zip(got, want):
            assert _same(g, w)


@pytest.mark.parametrize("N,P1,P2,normals", [(4, 2048, 1500, False), (2, 30000, 41000, True)])
def test_graph_chamfer_forward_backward(dev, N, P1, P2, normals):
    from pytorch3d_pointops_amd import graphs
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

    x, y = _cloud(7, (N, P1, 3), dev).requires_grad_(True), _cloud(8, (N, P2, 3), dev).requires_grad_(True)
    lx = torch.randint(P1 // 2, P1 + 1, (N,), generator=torch.Generator().manual_seed(9)).to(dev)
    ly = torch.randint(P2 // 2, P2 + 1, (N,), generator=torch.Generator().manual_seed(10)).to(dev)
    ins = [x, y]
    if normals:
        ins += [(_cloud(11, (N, P1, 3), dev) - 0.5).requires_grad_(True), (_cloud(12, (N, P2, 3), dev) - 0.5).requires_grad_(True)]

    def fn(*t):
        kw = dict(x_features={"normals": t[2]}, y_features={"normals": t[3]}, feature_names=["normals"]) if normals else {}
        loss, lf = chamfer_distance(t[0], t[1], x_lengths=lx, y_lengths=ly, **kw)
        total = loss + (0.25 * lf["normals"] if normals else 0.0)
        return total, loss

    step = graphs.capture(fn, ins, backward=True)

    def eager(ts):
        ts = [t.detach().clone().requires_grad_(True) for t in ts]
        total, loss = fn(*ts)
        return (total.detach(), loss.detach()), torch.autograd.grad(total, ts)

    fresh = [(_cloud(20 + i, tuple(t.shape), dev) - (0.5 if i >= 2 else 0.0)) for i, t in enumerate(ins)]
    for data in ([t.detach().clone() for t in ins], fresh):
        (wt, wl), wg = eager(data)
        (gt, gl), gg = step(*data)
        assert np.allclose(gt.item(), wt.item(), rtol=1e-5) and np.allclose(gl.item(), wl.item(), rtol=1e-5)
        for u, v in zip(gg, wg):
            u, v = u.cpu().numpy(), v.cpu().numpy()
            tol = 2e-5 * np.abs(v) + 2e-6 * max(1e-3, float(np.abs(v).max()))
            assert (np.abs(u - v) <= tol).all()


def test_graph_bypasses_grid_reuse(dev):
    import pytorch3d_pointops_amd as pa
    from pytorch3d_pointops_amd import _C, graphs
    from pytorch3d_pointops_amd.functions import knn_points

    a, b = _cloud(13, (2, 20000, 3), dev), _cloud(14, (2, 20000, 3), dev)
    pa.set_grid_cache(True)
    try:
        fn = lambda p, q: knn_points(p, q, K=8).idx  # noqa: E731
        fn(a, b)
        before = dict(_C.grid_cache_stats)
        step = graphs.capture(fn, (a, b), warmup=1)
        assert _C.grid_cache_stats["miss"] + _C.grid_cache_stats["points"] + _C.grid_cache_stats["both"] \
            == before["miss"] + before["points"] + before["both"] + 1  # the warm-up call only, not the captured one
        b2 = _cloud(15, (2, 20000, 3), dev)
        with torch.no_grad():
            b.data.copy_(b2)  # a write the cache cannot see: a replay must search the NEW points
        assert _same(step(), knn_points(a, b2, K=8).idx)
    finally:
        pa.set_grid_cache(False)


def test_graph_rejects_cpu_and_shape_changes(dev):
    from pytorch3d_pointops_amd import graphs
    from pytorch3d_pointops_amd.functions import knn_points

    a = _cloud(16, (1, 500, 3), dev)
    with pytest.raises(RuntimeError):
        graphs.capture(lambda p: knn_points(p, p, K=4).idx, (a.cpu(),))
    step = graphs.capture(lambda p: knn_points(p, p, K=4).idx, (a,))
    with pytest.raises(RuntimeError):
        step(_cloud(17, (1, 501, 3), dev))
