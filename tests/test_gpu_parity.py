"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
public functions.* API -> ctypes -> C ABI, against (a) the golden vectors the
reference itself produced and (b) the CPU oracle on the same seeded inputs.

Bars: idx outputs bit-exact; fp32 distance outputs bit-exact where the arithmetic is
order-free (knn / ball query dists, grad_p1, gathers, copies), 1e-5 where fp32 atomics
or tree reductions reorder sums (grad_p2, chamfer losses).
"""
import hashlib
import json
import os
import time

import numpy as np
import pytest
import torch

import cases
from conftest import GOLDEN, bits, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star tolerance for fp32 dists / chamfer


def G(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def close(a, b, tol=TOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    return a.shape == b.shape and (a.size == 0 or float(np.abs(a - b).max()) <= tol * scale)


def test_native_library_is_loaded(dev):
    from pytorch3d_pointops_amd import _C

    maps = open("/proc/self/maps").read()
    assert "libpointops_amd.so" in maps
    # exactly one HIP runtime in the process (ours must bind to the one torch loaded)
    hips = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(hips) == 1, hips
    assert _C.knn_check_version(3, 3, 16)


# ------------------------------------------------------------------ KNN
@pytest.mark.parametrize("name", sorted(cases.knn_cases()))
def test_knn_points(dev, oracle, name):
    from pytorch3d_pointops_amd.functions import knn_points

    g = load_golden("knn")
    c = cases.knn_cases()[name]
    r = knn_points(G(c["p1"], dev), G(c["p2"], dev), G(c["l1"], dev), G(c["l2"], dev), norm=c["norm"], K=c["K"],
                   return_nn=True)
    idx = r.idx.cpu().numpy()
    d = r.dists.cpu().numpy()
    assert np.array_equal(idx, g[name + "/idx"].astype(np.int64)), "idx differs from the reference golden"
    assert np.array_equal(bits(d), bits(g[name + "/dists"])), "dists differ from the reference golden"
    assert np.array_equal(r.knn.cpu().numpy(), g[name + "/knn"])
    oi, od = oracle.knn_points_idx(c["p1"], c["p2"], c["l1"], c["l2"], c["norm"], c["K"])
    assert np.array_equal(idx, oi) and np.array_equal(bits(d), bits(od))


@pytest.mark.parametrize("version", [-1, 0, 1, 2, 3])
def test_knn_versions_agree(dev, oracle, version):
    """`version` never changes results (reference: functions/knn.py:121, knn.cu:351-365)."""
    from pytorch3d_pointops_amd.functions import knn_points

    c = cases.knn_cases()["ties_lattice_k16"]
    r = knn_points(G(c["p1"], dev), G(c["p2"], dev), G(c["l1"], dev), G(c["l2"], dev), K=c["K"], version=version)
    oi, od = oracle.knn_points_idx(c["p1"], c["p2"], c["l1"], c["l2"], 2, c["K"])
    assert np.array_equal(r.idx.cpu().numpy(), oi)
    assert np.array_equal(bits(r.dists.cpu().numpy()), bits(od))


@pytest.mark.parametrize("name", sorted(n for n, c in cases.knn_cases().items() if c["p1"].shape[2] <= 3 and c["K"] <= 32))
def test_knn_grid_version_matches_golden(dev, name):
    """version=3 forces the exact grid-pruned search (+ brute-force fallback) on every small case."""
    from pytorch3d_pointops_amd.functions import knn_points

    g = load_golden("knn")
    c = cases.knn_cases()[name]
    r = knn_points(G(c["p1"], dev), G(c["p2"], dev), G(c["l1"], dev), G(c["l2"], dev), norm=c["norm"], K=c["K"],
                   version=3)
    assert np.array_equal(r.idx.cpu().numpy(), g[name + "/idx"].astype(np.int64))
    assert np.array_equal(bits(r.dists.cpu().numpy()), bits(g[name + "/dists"]))


def _grid_adversarial_cases():
    c = {}
    n = 3000
    base = cases.cloud(1301, (2, n, 3))
    c["disjoint_far"] = (base + np.float32(5.0), base, 8)           # every query outside p2's box -> fallback
    c["clustered"] = ((cases.cloud(1302, (2, n, 3)) ** np.float32(6.0)).astype(np.float32),
                      (cases.cloud(1303, (2, n, 3)) ** np.float32(6.0)).astype(np.float32), 16)
    flat = cases.cloud(1304, (2, n, 3)); flat[..., 2] = np.float32(0.25)
    c["planar"] = (flat, flat, 8)
    line = cases.cloud(1305, (2, n, 3)); line[..., 1:] = np.float32(-1.0)
    c["collinear"] = (line, line, 4)
    same = np.zeros((2, n, 3), np.float32) + np.float32(0.7)
    c["all_identical"] = (same, same, 5)
    c["lattice_ties"] = (cases.lattice(1306, 2, n, levels=6), cases.lattice(1307, 2, n, levels=6), 16)
    c["offset_1e3"] = (cases.cloud(1308, (2, n, 3)) + np.float32(1000.0), cases.cloud(1309, (2, n, 3)) + np.float32(1000.0), 8)
    c["tiny_scale"] = (cases.cloud(1310, (2, n, 3)) * np.float32(1e-20), cases.cloud(1311, (2, n, 3)) * np.float32(1e-20), 8)
    c["k1"] = (cases.cloud(1312, (2, n, 3)), cases.cloud(1313, (2, 2 * n, 3)), 1)
    c["k32"] = (cases.cloud(1314, (2, n, 3)), cases.cloud(1315, (2, n, 3)), 32)
    c["k11_l1"] = (cases.cloud(1316, (2, n, 3)), cases.cloud(1317, (2, n, 3)), 11)
    c["d2"] = (cases.cloud(1318, (2, n, 2)), cases.cloud(1319, (2, n, 2)), 8)
    c["d1"] = (cases.cloud(1320, (2, n, 1)), cases.cloud(1321, (2, n, 1)), 8)
    # half of each cloud inside a tiny cluster: a few grid blocks hold thousands of queries and
    # candidates (the chunk-level work split keeps this from serialising on one wave)
    m = 20000
    a = cases.cloud(1322, (2, m, 3)); a[:, : m // 2] = a[:, : m // 2] * np.float32(1e-3) + np.float32(0.5)
    b = cases.cloud(1323, (2, m, 3)); b[:, : m // 2] = b[:, : m // 2] * np.float32(1e-3) + np.float32(0.5)
    c["half_in_cluster"] = (a, b, 8)
    return c


_WIDE = {  # name: (N, P1, P2, D, K, norm, l1, l2)
    "d64_k20": (2, 300, 700, 64, 20, 2, [300, 123], [700, 650]),
    "d33_k5_l1": (2, 260, 515, 33, 5, 1, [260, 1], [515, 3]),
    "d16_k64": (2, 130, 400, 16, 64, 2, [130, 77], [400, 50]),  # K > len2 for cloud 1
    "d3_k100": (3, 200, 1000, 3, 100, 2, [200, 0, 64], [1000, 900, 0]),
    "d9_k16": (1, 515, 1030, 9, 16, 2, [515], [1030]),
    "d3_k33": (2, 200, 3000, 3, 33, 2, [200, 150], [3000, 40]),
    "d70_k48_l1": (1, 100, 600, 70, 48, 1, [100], [600]),
    "d3_k64_lattice": (1, 150, 2000, 3, 64, 2, [150], [2000]),
    "d128_k32": (1, 70, 300, 128, 32, 2, [70], [300]),
    "d3_k300_lds_cap": (1, 64, 700, 3, 300, 2, [64], [700]),
    "d200_k4_generic": (1, 40, 90, 200, 4, 2, [40], [77]),  # beyond the LDS budget -> plain generic kernel
}


@pytest.mark.parametrize("name", sorted(_WIDE))
def test_knn_wide_shapes(dev, oracle, monkeypatch, name):
    """Feature-space D and long lists (knn_wide.hip: LDS-transposed queries, register or LDS lists),
    bit-exact against the oracle, and identical to the plain generic kernel."""
    from pytorch3d_pointops_amd import _C

    N, P1, P2, D, K, norm, l1, l2 = _WIDE[name]
    p1 = cases.cloud(1700 + D, (N, P1, D))
    p2 = cases.cloud(1701 + D + K, (N, P2, D))
    if name in ("d3_k100", "d3_k64_lattice"):
        p2 = cases.lattice(1702, N, P2, D, levels=5)  # many exact ties inside long lists
        p1 = cases.lattice(1703, N, P1, D, levels=5)
    l1, l2 = np.array(l1), np.array(l2)
    idx, dists = _C.knn_points_idx(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm, K, -1)
    oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(dists.cpu().numpy()), bits(od))
    monkeypatch.setenv("POINTOPS_DEBUG", "knn_generic=1")
    idx2, dists2 = _C.knn_points_idx(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm, K, 0)
    assert torch.equal(idx, idx2) and torch.equal(dists.view(torch.int32), dists2.view(torch.int32))


_SMALL = {  # name: (N, P1, P2, D, K, norm, l1, l2)
    "cfg1": (2, 1024, 1024, 3, 8, 2, [1024, 1024], [1024, 1024]),
    "ragged_k_gt_len2": (3, 130, 700, 3, 16, 2, [130, 0, 77], [700, 5, 0]),
    "k1_l1": (2, 200, 333, 3, 1, 1, [200, 199], [333, 1]),
    "k32_d8": (2, 65, 515, 8, 32, 2, [65, 64], [515, 40]),
    "k24_d5": (1, 100, 1000, 5, 24, 1, [100], [999]),
    "k2_d1": (2, 300, 257, 1, 2, 2, [300, 1], [257, 256]),
    "k4_d2_lattice": (2, 256, 900, 2, 4, 2, [256, 100], [900, 899]),
    "k16_lattice": (2, 128, 2000, 3, 16, 2, [128, 128], [2000, 64]),
    "few_queries_long_cloud": (1, 7, 20000, 3, 8, 2, [6], [20000]),
    "many_queries_per_wave": (1, 40000, 100, 3, 4, 2, [39999], [100]),  # 3 queries per wave
    "identical_points": (1, 70, 300, 3, 8, 2, [70], [300]),
    # long clouds: the wave-uniform gates are refreshed several times; on a lattice most candidates TIE with the gate
    "k32_long_cloud_gates": (1, 5, 20000, 3, 32, 2, [5], [19999]),
    "k4_long_lattice_gate_ties": (2, 40, 9000, 3, 4, 2, [40, 13], [9000, 8999]),
    "k1_long_lattice_gate_ties_l1": (1, 70, 6000, 2, 1, 1, [70], [6000]),
    "k8_identical_long": (1, 9, 5000, 3, 8, 2, [9], [5000]),
}


@pytest.mark.parametrize("per_wave", ["one", "shared"])
@pytest.mark.parametrize("name", sorted(_SMALL))
def test_knn_small_batches(dev, oracle, monkeypatch, name, per_wave):
    """Few queries (knn_small.hip: one wave per query, the cloud dealt over the lanes, answers pulled out of the 64
    list heads by wave-wide minima), forced on: bit-exact against the oracle -- ragged lengths, K > len2, ties
    (lattices, identical points: the smaller index wins), D = 1..8, both norms -- and identical to the sliced
    lane-per-query scan it replaces at these sizes.  "shared": up to four queries per wave share each loaded
    candidate (ragged query counts leave partial groups); the long cloud exercises the wave-uniform gates."""
    from pytorch3d_pointops_amd import _C

    N, P1, P2, D, K, norm, l1, l2 = _SMALL[name]
    p1 = cases.cloud(1900 + D, (N, P1, D))
    p2 = cases.cloud(1901 + D + K, (N, P2, D))
    if "lattice" in name:
        p1, p2 = cases.lattice(1902, N, P1, D, levels=5), cases.lattice(1903, N, P2, D, levels=5)
    if name in ("identical_points", "k8_identical_long"):
        p2[:] = np.float32(0.25)
    l1, l2 = np.array(l1), np.array(l2)
    monkeypatch.setenv("POINTOPS_DEBUG", "knn_small=1,knn_small_q=" + ("1" if per_wave == "one" else "2"))
    idx, dists = _C.knn_points_idx(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm, K, 2)
    oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(dists.cpu().numpy()), bits(od))
    monkeypatch.setenv("POINTOPS_DEBUG", "knn_small=0")
    idx2, dists2 = _C.knn_points_idx(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm, K, 2)
    assert torch.equal(idx, idx2) and torch.equal(dists.view(torch.int32), dists2.view(torch.int32))


@pytest.mark.parametrize("name", sorted(_grid_adversarial_cases()))
def test_knn_grid_adversarial(dev, oracle, name):
    """Grid search vs the CPU oracle on distributions that stress the bound / fallback logic,
    with ragged lengths (one cloud shorter than K for the second query cloud)."""
    from pytorch3d_pointops_amd.functions import knn_points

    p1, p2, K = _grid_adversarial_cases()[name]
    norm = 1 if name.endswith("_l1") else 2
    l1 = np.array([p1.shape[1], p1.shape[1] // 3])
    l2 = np.array([p2.shape[1], max(K - 2, 1)])
    r = knn_points(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm=norm, K=K, version=3)
    oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
    assert np.array_equal(r.idx.cpu().numpy(), oi)
    assert np.array_equal(bits(r.dists.cpu().numpy()), bits(od))
    # and the brute-force family agrees too
    r2 = knn_points(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm=norm, K=K, version=2)
    assert torch.equal(r2.idx, r.idx) and torch.equal(r2.dists, r.dists)


@pytest.mark.parametrize("mode", ["quad", "noquad", "self"])
@pytest.mark.parametrize("name", ["clustered", "lattice_ties", "disjoint_far", "half_in_cluster", "k32", "k1", "d2"])
def test_knn_grid_alternative_passes(dev, oracle, monkeypatch, mode, name):
    """The passes the automatic choice only takes on large clouds (the radius-2 quad search needs >= 32768
    queries per cloud), forced on the adversarial distributions; "self": the queries ARE the points (same
    buffer), so the point sort doubles as the query order -- against the run with that reuse switched off."""
    from pytorch3d_pointops_amd.functions import knn_points

    p1, p2, K = _grid_adversarial_cases()[name]
    if mode == "self":
        l2 = np.array([p2.shape[1], max(K - 2, 1)])
        t, lt = G(p2, dev), G(l2, dev)
        r = knn_points(t, t, lt, lt, K=K, version=3)  # same tensors: one sort
        oi, od = oracle.knn_points_idx(p2, p2, l2, l2, 2, K)
        assert np.array_equal(r.idx.cpu().numpy(), oi)
        assert np.array_equal(bits(r.dists.cpu().numpy()), bits(od))
        monkeypatch.setenv("POINTOPS_DEBUG", "grid_same=0")
        r2 = knn_points(t, t, lt, lt, K=K, version=3)
        assert torch.equal(r2.idx, r.idx) and torch.equal(r2.dists, r.dists)
        return
    monkeypatch.setenv("POINTOPS_DEBUG", "grid_quad=1" if mode == "quad" else "grid_quad=0")
    l1 = np.array([p1.shape[1], p1.shape[1] // 3])
    l2 = np.array([p2.shape[1], max(K - 2, 1)])
    r = knn_points(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), K=K, version=3)
    oi, od = oracle.knn_points_idx(p1, p2, l1, l2, 2, K)
    assert np.array_equal(r.idx.cpu().numpy(), oi)
    assert np.array_equal(bits(r.dists.cpu().numpy()), bits(od))


def test_knn_default_lengths_and_empty(dev, oracle):
    from pytorch3d_pointops_amd.functions import knn_points

    p1 = cases.cloud(901, (2, 77, 3))
    p2 = cases.cloud(902, (2, 301, 3))
    r = knn_points(G(p1, dev), G(p2, dev), K=7)
    oi, od = oracle.knn_points_idx(p1, p2, np.array([77, 77]), np.array([301, 301]), 2, 7)
    assert np.array_equal(r.idx.cpu().numpy(), oi) and np.array_equal(bits(r.dists.cpu().numpy()), bits(od))
    assert r.knn is None
    # P1 == 0 works and returns (N,0,K)
    e = knn_points(torch.zeros(2, 0, 3, device=dev), G(p2, dev), K=3)
    assert tuple(e.idx.shape) == (2, 0, 3) and tuple(e.dists.shape) == (2, 0, 3)
    with pytest.raises(ValueError, match="1 or 2 norm"):
        knn_points(G(p1, dev), G(p2, dev), norm=3)
    # non-contiguous inputs are made contiguous (functions/knn.py:178-179)
    pt = G(np.ascontiguousarray(p1.transpose(0, 2, 1)), dev).transpose(1, 2)
    r2 = knn_points(pt, G(p2, dev), K=7)
    assert torch.equal(r2.idx, r.idx)


@pytest.mark.parametrize("name", sorted(cases.knn_backward_cases()))
def test_knn_backward(dev, oracle, name):
    from pytorch3d_pointops_amd.functions import knn_points

    g = load_golden("knn_backward")
    c = cases.knn_backward_cases()[name]
    p1 = G(c["p1"], dev).requires_grad_(True)
    p2 = G(c["p2"], dev).requires_grad_(True)
    r = knn_points(p1, p2, G(c["l1"], dev), G(c["l2"], dev), norm=c["norm"], K=c["K"])
    assert not r.idx.requires_grad
    grad = cases.grad_for(name, tuple(r.dists.shape))
    r.dists.backward(G(grad, dev))
    g1 = p1.grad.cpu().numpy()
    g2 = p2.grad.cpu().numpy()
    # grad_p1: per-query register sum in k order == CPU order -> bit-exact
    assert np.array_equal(bits(g1), bits(g[name + "/grad_p1"]))
    # grad_p2: fp32 atomics, order-dependent last bits
    assert close(g2, g[name + "/grad_p2"])
    o1, o2 = oracle.knn_points_backward(c["p1"], c["p2"], c["l1"], c["l2"], r.idx.cpu().numpy(), c["norm"], grad)
    assert np.array_equal(bits(g1), bits(o1)) and close(g2, o2)


@pytest.mark.parametrize("mode,split", [("tiled", None), ("tiled", "3"), ("atomic", None)])
@pytest.mark.parametrize("D,norm,K", [(3, 2, 8), (3, 1, 8), (2, 2, 8), (4, 2, 8), (1, 2, 8), (3, 2, 1), (3, 2, 3), (3, 2, 21)])
def test_knn_backward_modes(dev, oracle, monkeypatch, mode, split, D, norm, K):
    """grad_p2 through the LDS-tile kernel (several tiles per cloud, ragged clouds, an empty cloud,
    row splits that meet with atomics) and through the device-atomic kernel: both against the
    oracle's CPU loop (knn_cpu.cpp:75-128) on the SAME neighbour table."""
    from pytorch3d_pointops_amd import _C

    monkeypatch.setenv("POINTOPS_DEBUG", f"knn_bwd_mode={mode}" + (f",knn_bwd_split={split}" if split else ""))
    N, P1, P2 = 3, 2500, 30000
    p1 = cases.cloud(1500 + D, (N, P1, D))
    p2 = cases.cloud(1510 + D, (N, P2, D))
    l1 = np.array([P1, 1777, 0])
    l2 = np.array([P2, 9001, 5])
    idx, _ = _C.knn_points_idx(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm, K, -1)
    grad = cases.grad_for("bwd_modes", (N, P1, K))
    g1, g2 = _C.knn_points_backward(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), idx, norm, G(grad, dev))
    o1, o2 = oracle.knn_points_backward(p1, p2, l1, l2, idx.cpu().numpy(), norm, grad)
    assert np.array_equal(bits(g1.cpu().numpy()), bits(o1))
    assert close(g2.cpu().numpy(), o2)
    # ball-query style table: -1 padding and out-of-tile rows are ignored alike
    idx2 = idx.clone()
    idx2[:, ::3, 1::2] = -1
    g1b, g2b = _C.knn_points_backward(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), idx2, norm, G(grad, dev))
    o1b, o2b = oracle.knn_points_backward(p1, p2, l1, l2, np.where(idx2.cpu().numpy() < 0, 0, idx2.cpu().numpy()), norm,
                                          np.where(idx2.cpu().numpy() < 0, 0.0, grad).astype(np.float32))
    assert close(g1b.cpu().numpy(), o1b) and close(g2b.cpu().numpy(), o2b)


@pytest.mark.parametrize("K", [6, 8])
@pytest.mark.parametrize("U", [1, 2, 3, 4, 7, 64])
def test_knn_gather_widths(dev, U, K):
    """knn_gather forward for every kernel variant (row kernels U <= 4, element kernel above) against the numpy restatement of functions/knn.py:200-250, with -1 rows
    and k >= lengths masking."""
    from oracle import oracle as O
    from pytorch3d_pointops_amd import _C, synth

    N, M, L = 3, 500, 333
    x = cases.cloud(1900 + U, (N, M, U))
    idx = synth.randint(1901, 0, M - 1, (N, L, K))
    lengths = np.array([K, 2, 0])
    out = _C.gather_neighbors(G(x, dev), G(idx, dev), G(lengths, dev))
    assert np.array_equal(out.cpu().numpy(), O.knn_gather(x, idx, lengths))
    idx2 = idx.copy()
    idx2[:, ::5, 2] = -1
    out2 = _C.gather_neighbors(G(x, dev), G(idx2, dev), None)
    assert np.array_equal(out2.cpu().numpy(), O.masked_gather(x, idx2))


@pytest.mark.parametrize("mode,split", [("tiled", None), ("tiled", "2"), ("atomic", None)])
@pytest.mark.parametrize("U", [1, 3, 4])
def test_gather_backward_modes(dev, monkeypatch, mode, split, U):
    """knn_gather / masked_gather backward (scatter-add of grad_out rows into grad_x) through the
    LDS-tile kernel and the device-atomic kernel against a float64 np.add.at of the same masks
    (functions/knn.py:236-248: k >= lengths[n] zeroed; utils.py:53-63: -1 rows zeroed)."""
    from pytorch3d_pointops_amd import _C, synth

    monkeypatch.setenv("POINTOPS_DEBUG", f"gather_bwd_mode={mode}" + (f",gather_bwd_split={split}" if split else ""))
    N, L, K, M = 2, 3000, 8, 20000
    idx = synth.randint(1601, -1, M - 1, (N, L, K))
    idx[0, ::7, :] = 3  # many rows onto one target row
    go = cases.grad_for("gbm%d" % U, (N, L, K, U))
    go[1, 5::11] = 0.0
    for lengths in (None, np.array([8, 5])):
        gx = _C.gather_neighbors_backward(G(go, dev), G(idx, dev), None if lengths is None else G(lengths, dev), M)
        ref = np.zeros((N, M, U), np.float64)
        for n in range(N):
            kk = K if lengths is None else int(lengths[n])
            ii = idx[n, :, :kk].reshape(-1)
            vv = go[n, :, :kk].reshape(-1, U).astype(np.float64)
            ok = ii >= 0
            np.add.at(ref[n], ii[ok], vv[ok])
        assert close(gx.cpu().numpy(), ref)


# ------------------------------------------------------------------ gather
def test_knn_gather_and_masked_gather(dev):
    from pytorch3d_pointops_amd.functions import knn_gather, masked_gather
    from pytorch3d_pointops_amd import synth

    g = load_golden("gather")
    x = cases.cloud(601, (2, 50, 5))
    idx = synth.randint(602, 0, 49, (2, 30, 4))
    lengths = np.array([50, 2])
    xr = G(x, dev).requires_grad_(True)
    o = knn_gather(xr, G(idx, dev), G(lengths, dev))
    assert np.array_equal(o.detach().cpu().numpy(), g["knn_gather/out"])
    up = cases.grad_for("kg", tuple(o.shape))
    (o * G(up, dev)).sum().backward()
    assert close(xr.grad.cpu().numpy(), g["knn_gather/grad_x"])
    midx = idx.copy()
    midx[0, ::3, 1] = -1
    midx[1, :, 3] = -1
    xr2 = G(x, dev).requires_grad_(True)
    o2 = masked_gather(xr2, G(midx, dev))
    assert np.array_equal(o2.detach().cpu().numpy(), g["masked_gather3/out"])
    (o2 * G(up, dev)).sum().backward()
    assert close(xr2.grad.cpu().numpy(), g["masked_gather3/grad_x"])
    o3 = masked_gather(G(x, dev), G(np.ascontiguousarray(midx[:, :, 1]), dev))
    assert np.array_equal(o3.cpu().numpy(), g["masked_gather2/out"])
    with pytest.raises(ValueError, match="not supported"):
        masked_gather(G(x, dev), G(midx, dev)[..., None])


# ------------------------------------------------------------------ ball query
@pytest.mark.parametrize("kernel", ["lane_per_query", "wave_per_query"])
@pytest.mark.parametrize("name", sorted(cases.ball_query_cases()))
def test_ball_query(dev, oracle, monkeypatch, name, kernel):
    """Reference goldens and the oracle, through the lane-per-query scan (ball_query.hip) and through the
    wave-per-query kernel that small batches take (ball_small.hip)."""
    from pytorch3d_pointops_amd.functions import ball_query

    monkeypatch.setenv("POINTOPS_DEBUG", "ball_small=" + ("1" if kernel == "wave_per_query" else "0"))
    g = load_golden("ball_query")
    c = cases.ball_query_cases()[name]
    p1 = G(c["p1"], dev).requires_grad_(True)
    p2 = G(c["p2"], dev).requires_grad_(True)
    r = ball_query(p1, p2, G(c["l1"], dev), G(c["l2"], dev), K=c["K"], radius=c["radius"], return_nn=True)
    idx = r.idx.cpu().numpy()
    d = r.dists.detach().cpu().numpy()
    assert np.array_equal(idx, g[name + "/idx"].astype(np.int64))
    assert np.array_equal(bits(d), bits(g[name + "/dists"]))
    assert np.array_equal(r.knn.detach().cpu().numpy(), g[name + "/knn"])
    oi, od = oracle.ball_query(c["p1"], c["p2"], c["l1"], c["l2"], c["K"], c["radius"])
    assert np.array_equal(idx, oi) and np.array_equal(bits(d), bits(od))
    assert (d[idx >= 0] < np.float32(c["radius"]) ** 2).all()  # the reference example's own check
    up = cases.grad_for("bq" + name, tuple(d.shape))
    (r.dists * G(up, dev)).sum().backward()
    assert np.array_equal(bits(p1.grad.cpu().numpy()), bits(g[name + "/grad_p1"]))
    assert close(p2.grad.cpu().numpy(), g[name + "/grad_p2"])


@pytest.mark.parametrize("kernel", ["lane_per_query", "wave_per_query"])
@pytest.mark.parametrize("radius,K,D", [(0.05, 16, 3), (0.3, 8, 3), (0.12, 64, 3), (0.3, 500, 3), (0.2, 20, 2), (0.6, 33, 6)])
def test_ball_query_larger_clouds(dev, oracle, monkeypatch, radius, K, D, kernel):
    """Larger clouds: sparse balls (queries that never fill scan everything), dense balls (early
    exit), K = 64 and the reference's default K = 500, ragged lengths, D = 2 / 3 and the run-time-D path (6);
    both kernels."""
    from pytorch3d_pointops_amd import _C

    monkeypatch.setenv("POINTOPS_DEBUG", "ball_small=" + ("1" if kernel == "wave_per_query" else "0"))
    p1 = cases.cloud(1501, (2, 3000, D))
    p2 = cases.cloud(1502, (2, 20000, D))
    l1 = np.array([3000, 1234])
    l2 = np.array([20000, 6000])
    idx, d = _C.ball_query(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), K, radius)
    oi, od = oracle.ball_query(p1, p2, l1, l2, K, radius)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))


@pytest.mark.parametrize("name", sorted(cases.ball_query_cases()))
def test_ball_query_grid_matches_golden(dev, monkeypatch, name):
    """The cell-grid path (forced on: these clouds are far below its automatic size threshold) gives
    the reference's results bit for bit, including the strict-< lattice boundary and D = 2."""
    from pytorch3d_pointops_amd import _C

    g = load_golden("ball_query")
    c = cases.ball_query_cases()[name]
    monkeypatch.setenv("POINTOPS_DEBUG", "ball_grid=1")
    idx, d = _C.ball_query(G(c["p1"], dev), G(c["p2"], dev), G(c["l1"], dev), G(c["l2"], dev), c["K"], c["radius"])
    assert np.array_equal(idx.cpu().numpy(), g[name + "/idx"].astype(np.int64))
    assert np.array_equal(bits(d.cpu().numpy()), bits(g[name + "/dists"]))


@pytest.mark.parametrize("radius,K,D", [(0.05, 16, 3), (0.02, 32, 3), (0.3, 8, 3), (0.12, 64, 3), (1e-4, 4, 3),
                                        (0.01, 8, 2), (0.001, 5, 1), (0.05, 100, 3), (-0.05, 16, 3)])
def test_ball_query_grid_vs_oracle(dev, oracle, monkeypatch, radius, K, D):
    """Grid path on larger ragged clouds (sparse balls -> grid, dense balls -> the device picks the
    scan, K = 100 -> no grid, clustered and offset data, an empty cloud), against the oracle and
    against the scan-only path."""
    from pytorch3d_pointops_amd import _C

    p1 = cases.cloud(1801, (3, 3000, D))
    p2 = cases.cloud(1802, (3, 20000, D))
    p2[1] = (p2[1] ** np.float32(3.0)).astype(np.float32) + np.float32(10.0)  # clustered, far from the origin
    p1[1] = (p1[1] ** np.float32(3.0)).astype(np.float32) + np.float32(10.0)
    l1 = np.array([3000, 1234, 77])
    l2 = np.array([20000, 6000, 0])
    monkeypatch.setenv("POINTOPS_DEBUG", "ball_grid=1")
    idx, d = _C.ball_query(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), K, radius)
    oi, od = oracle.ball_query(p1, p2, l1, l2, K, radius)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))
    monkeypatch.setenv("POINTOPS_DEBUG", "ball_grid=0")
    idx0, d0 = _C.ball_query(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), K, radius)
    assert torch.equal(idx, idx0) and torch.equal(d.view(torch.int32), d0.view(torch.int32))


@pytest.mark.parametrize("name", ["clustered", "planar", "collinear", "all_identical", "lattice_ties", "offset_1e3",
                                  "tiny_scale", "disjoint_far", "half_in_cluster", "d2", "d1"])
def test_ball_query_grid_adversarial(dev, oracle, monkeypatch, name):
    """Grid path of the ball query forced on (also past the device-side density test) for degenerate and
    skewed distributions: flat / collinear / identical points, lattices whose distances sit exactly on
    radius^2, offsets, tiny scales, disjoint clouds, dense clusters."""
    from pytorch3d_pointops_amd import _C

    p1, p2, _ = _grid_adversarial_cases()[name]
    scale = float(np.abs(p2).max()) if name == "tiny_scale" else 1.0
    radius = 0.25 * scale if name == "lattice_ties" else 0.06 * scale
    l1 = np.array([p1.shape[1], p1.shape[1] // 3])
    l2 = np.array([p2.shape[1], 5])
    monkeypatch.setenv("POINTOPS_DEBUG", "ball_grid=1,ball_factor=0")  # always the grid where one can be built
    for K in (4, 20):
        idx, d = _C.ball_query(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), K, radius)
        oi, od = oracle.ball_query(p1, p2, l1, l2, K, radius)
        assert np.array_equal(idx.cpu().numpy(), oi), (name, K)
        assert np.array_equal(bits(d.cpu().numpy()), bits(od)), (name, K)


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("kernel", ["auto", "clusters"])
@pytest.mark.parametrize("name", sorted(cases.fps_cases()))
def test_sample_farthest_points(dev, oracle, monkeypatch, name, kernel):
    """Reference goldens + oracle; "auto": clouds of up to 4096 points take the four-wave kernel (fps_small_kernel),
    "clusters": the 16-wave cluster kernel they took before (POINTOPS_DEBUG fps_small=0)."""
    from pytorch3d_pointops_amd import _C

    if kernel == "clusters":
        monkeypatch.setenv("POINTOPS_DEBUG", "fps_small=0")
    from pytorch3d_pointops_amd.functions import masked_gather, sample_farthest_points
    from pytorch3d_pointops_amd.functions.sample_farthest_points import sample_farthest_points_naive

    g = load_golden("fps")
    c = cases.fps_cases()[name]
    pts = G(c["points"], dev)
    idx = _C.sample_farthest_points(pts, G(c["lengths"], dev), G(c["K"], dev), G(c["start"], dev))
    got = idx.cpu().numpy()
    assert np.array_equal(got, g[name + "/idx"].astype(np.int64))
    assert np.array_equal(got, oracle.sample_farthest_points(c["points"], c["lengths"], c["K"], c["start"]))
    assert np.array_equal(masked_gather(pts, idx).cpu().numpy(), g[name + "/points"])
    if (c["start"] == 0).all():
        sp, si = sample_farthest_points(pts, G(c["lengths"], dev), G(c["K"], dev))
        assert torch.equal(si, idx)
        assert np.array_equal(sp.cpu().numpy(), g[name + "/points"])
        if name not in ("all_equal", "big_cloud"):
            _, ni = sample_farthest_points_naive(pts, G(c["lengths"], dev), G(c["K"], dev))
            assert torch.equal(ni, idx)  # examples/fps_on_pointclouds.py:153


def test_fps_multi_workgroup_clusters(dev, oracle):
    """Clouds larger than one workgroup's register capacity are split over a cluster of
    workgroups that exchange their local argmax through per-iteration atomic slots; several
    ragged clouds share the clusters (persistent loop)."""
    from pytorch3d_pointops_amd import _C

    for (N, P, D) in ((5, 10000, 3), (3, 40000, 3), (3, 9000, 2), (5, 4096, 3), (4, 3000, 2), (5, 1024, 3), (3, 700, 2)):
        pts = cases.cloud(1400 + P, (N, P, D))
        pts[0, 100:200] = pts[0, 0:100]  # duplicates -> ties on the running min-distance
        lengths = np.array([P, P // 2 + 7, min(4097, P), 5, P - 1][:N])
        K = np.array([64, 300, 17, 9, 128][:N])
        start = np.array([0, 11, min(4096, P - 1), 4, P - 2][:N])
        got = _C.sample_farthest_points(G(pts, dev), G(lengths, dev), G(K, dev), G(start, dev)).cpu().numpy()
        want = oracle.sample_farthest_points(pts, lengths, K, start)
        assert np.array_equal(got, want), (N, P, D)


@pytest.mark.parametrize("knob", ["", "fps_small=0"])
def test_fps_zero_k_cloud_reports_its_start_index(dev, oracle, monkeypatch, knob):
    """A cloud with K[n] = 0 in a batch whose max K is positive: the reference's CPU kernel writes the start index
    into slot 0 BEFORE it looks at K (sample_farthest_points_cpu.cpp:53-57), so the row is [start, -1, ...]; an empty
    cloud stays all -1.  (Found by tests/test_fuzz_small_gpu.py: rounds 1-2 returned -1 there.)"""
    from pytorch3d_pointops_amd import _C

    monkeypatch.setenv("POINTOPS_DEBUG", knob)
    for P in (500, 6000, 20000):  # four-wave kernel / one cluster workgroup / several
        pts = cases.cloud(2300 + P, (4, P, 3))
        L, K, S = np.array([P, P - 7, 0, 33]), np.array([5, 0, 3, 1]), np.array([1, 9, 0, 32])
        got = _C.sample_farthest_points(G(pts, dev), G(L, dev), G(K, dev), G(S, dev)).cpu().numpy()
        assert np.array_equal(got, oracle.sample_farthest_points(pts, L, K, S))
        assert got[1].tolist() == [9, -1, -1, -1, -1] and got[2].tolist() == [-1] * 5 and got[3].tolist() == [32, -1, -1, -1, -1]


def test_fps_int_and_list_K_and_random_start(dev):
    from pytorch3d_pointops_amd.functions import sample_farthest_points

    pts = G(cases.cloud(301, (4, 500, 3)), dev)
    a, ai = sample_farthest_points(pts, K=32)
    g = load_golden("fps")
    assert np.array_equal(ai.cpu().numpy(), g["fixed_k/idx"].astype(np.int64))
    b, bi = sample_farthest_points(pts, lengths=G(np.array([500, 120, 33, 1]), dev), K=[10, 200, 5, 3])
    assert np.array_equal(bi.cpu().numpy(), g["per_cloud_k/idx"].astype(np.int64))
    torch.manual_seed(0)
    c, ci = sample_farthest_points(pts, K=8, random_start_point=True)
    assert tuple(ci.shape) == (4, 8) and (ci >= 0).all()
    # differentiable w.r.t. points through the gather
    pr = pts.clone().requires_grad_(True)
    sp, _ = sample_farthest_points(pr, K=4)
    sp.sum().backward()
    assert float(pr.grad.sum()) == 4 * 4 * 3


# ------------------------------------------------------------------ packed <-> padded
@pytest.mark.parametrize("name", sorted(cases.packed_cases()))
def test_packed_padded(dev, name):
    from pytorch3d_pointops_amd.functions import packed_to_padded, padded_to_packed

    g = load_golden("packed_padded")
    c = cases.packed_cases()[name]
    x, first, F = cases.packed_inputs(c)
    xt = G(x, dev).requires_grad_(True)
    arg = xt[:, 0] if c["D"] == 1 else xt
    padded = packed_to_padded(arg, G(first, dev), int(c["max_size"]))
    assert np.array_equal(padded.detach().cpu().numpy(), g[name + "/padded"])
    up = cases.grad_for("pp" + name, tuple(padded.shape))
    (padded * G(up, dev)).sum().backward()
    assert np.array_equal(xt.grad.cpu().numpy(), g[name + "/grad_packed"])
    back = padded_to_packed(padded.detach(), G(first, dev), F)
    assert np.array_equal(back.cpu().numpy(), g[name + "/roundtrip"])
    assert np.array_equal(back.cpu().numpy().reshape(x.shape if c["D"] > 1 else (F,)), x if c["D"] > 1 else x[:, 0])
    pt = G(cases.grad_for("pq" + name, tuple(padded.shape)), dev).requires_grad_(True)
    pk = padded_to_packed(pt, G(first, dev), F)
    assert np.array_equal(pk.detach().cpu().numpy(), g[name + "/packed_of_g"])
    pk.sum().backward()
    assert np.array_equal(pt.grad.cpu().numpy(), g[name + "/grad_padded_ones"])


def test_packed_padded_trailing_dims(dev):
    from pytorch3d_pointops_amd.functions import packed_to_padded, padded_to_packed

    g = load_golden("packed_padded")
    x = cases.cloud(410, (15, 2, 3))
    first = np.array([0, 5, 5, 12], np.int64)
    p3 = packed_to_padded(G(x, dev), G(first, dev), 7)
    assert np.array_equal(p3.cpu().numpy(), g["trailing/padded"])
    assert np.array_equal(padded_to_packed(p3, G(first, dev), 15).cpu().numpy(), g["trailing/roundtrip"])
    # max_size_dim != 1
    q = p3.movedim(1, 2).contiguous()
    assert np.array_equal(padded_to_packed(q, G(first, dev), 15, max_size_dim=2).cpu().numpy(), g["trailing/roundtrip"])


# ------------------------------------------------------------------ get_point_covariances / wmean
@pytest.mark.parametrize("name", ["d3_k8", "d2_k5", "d3_k16_ties"])
def test_get_point_covariances(dev, name):
    """Fused covariance kernel + closed-form backward against the reference's composed torch ops
    (functions/utils.py:111-153; goldens from the reference itself), values and gradients within 1e-5."""
    from pytorch3d_pointops_amd.functions.utils import get_point_covariances

    g = load_golden("covariances")
    N, P, D, K, lens = {"d3_k8": (2, 120, 3, 8, [120, 57]), "d2_k5": (1, 90, 2, 5, [90]),
                        "d3_k16_ties": (1, 150, 3, 16, [150])}[name]
    pts = cases.lattice(811, N, P, D, levels=6) if "ties" in name else cases.cloud(810 + D + K, (N, P, D))
    x = G(pts, dev).requires_grad_(True)
    cov, knn = get_point_covariances(x, G(np.array(lens), dev), K)
    assert np.array_equal(knn.detach().cpu().numpy(), g[name + "/knn"])
    assert close(cov.detach().cpu().numpy(), g[name + "/cov"])
    gc = G(cases.grad_for("cov" + name, tuple(cov.shape)), dev)
    gk = G(cases.grad_for("knn" + name, tuple(knn.shape)), dev)
    ((cov * gc).sum() + (knn * gk).sum()).backward()
    assert close(x.grad.cpu().numpy(), g[name + "/grad_points"])


def test_point_covariances_wide_and_wmean(dev):
    """D up to 8 (runtime-D kernel) against the composed torch expression on the same neighbourhoods;
    wmean against the reference's values."""
    from pytorch3d_pointops_amd import _C
    from pytorch3d_pointops_amd.functions.utils import wmean

    for D, K in ((1, 3), (5, 7), (8, 12)):
        knn = G(cases.cloud(2100 + D, (2, 300, K, D)), dev).requires_grad_(True)
        m = knn.mean(2, keepdim=True)
        cd = knn - m
        ref = (cd.unsqueeze(4) * cd.unsqueeze(3)).mean(2)
        gc = G(cases.grad_for("covw%d" % D, tuple(ref.shape)), dev)
        (ref * gc).sum().backward()
        cov = _C.point_covariances(knn.detach())
        gk = _C.point_covariances_backward(knn.detach(), gc)
        assert close(cov.cpu().numpy(), ref.detach().cpu().numpy())
        assert close(gk.cpu().numpy(), knn.grad.cpu().numpy())
    g = load_golden("covariances")
    from pytorch3d_pointops_amd import synth
    xw = G(cases.cloud(820, (2, 40, 3)), dev)
    w = G(synth.uniform_f32(821, (2, 40)), dev)
    assert close(wmean(xw, w).cpu().numpy(), g["wmean/weighted"])
    assert close(wmean(xw).cpu().numpy(), g["wmean/plain"])


# ------------------------------------------------------------------ chamfer
def _chamfer_call(dev, inp, v, as_leaf=True):
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

    x = G(inp["x"], dev).requires_grad_(as_leaf)
    y = G(inp["y"], dev).requires_grad_(as_leaf)
    xn = G(inp["xn"], dev).requires_grad_(as_leaf)
    yn = G(inp["yn"], dev).requires_grad_(as_leaf)
    kw = dict(x_lengths=G(inp["xl"], dev), y_lengths=G(inp["yl"], dev), batch_reduction=v["batch_reduction"],
              point_reduction=v["point_reduction"], norm=v["norm"], single_directional=v["single_directional"],
              abs_cosine=v["abs_cosine"])
    if v["use_weights"]:
        kw["weights"] = G(inp["w"], dev)
    if v["features"]:
        kw.update(x_features={"normals": xn}, y_features={"normals": yn}, feature_names=["normals"])
    loss, lf = chamfer_distance(x, y, **kw)
    return x, y, xn, yn, loss, lf


@pytest.mark.parametrize("v", cases.chamfer_variants(), ids=cases.variant_key)
def test_chamfer_distance(dev, v):
    g = load_golden("chamfer")
    key = cases.variant_key(v)
    inp = cases.chamfer_inputs()
    x, y, xn, yn, loss, lf = _chamfer_call(dev, inp, v)
    flat = []

    def chk(tag, t):
        if isinstance(t, tuple):
            for i, tt in enumerate(t):
                chk(f"{tag}{i}", tt)
        elif t is not None:
            want = g[f"{key}/{tag}"]
            assert close(t.detach().cpu().numpy(), want), (key, tag)
            flat.append(t)

    chk("loss", loss)
    if v["features"]:
        assert lf is not None
        chk("lossf", lf["normals"])
    else:
        assert lf is None
    sum(t.sum() for t in flat).backward()
    assert close(x.grad.cpu().numpy(), g[f"{key}/grad_x"]), key
    assert close(y.grad.cpu().numpy(), g[f"{key}/grad_y"]), key
    if v["features"]:
        assert close(xn.grad.cpu().numpy(), g[f"{key}/grad_xn"]), key
        assert close(yn.grad.cpu().numpy(), g[f"{key}/grad_yn"]), key


def test_chamfer_pointclouds_input(dev):
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance
    from pytorch3d_pointops_amd.structures import Pointclouds

    g = load_golden("chamfer")
    inp = cases.chamfer_inputs()
    xl, yl = inp["xl"], inp["yl"]
    pc_x = Pointclouds([G(inp["x"][n, : xl[n]], dev) for n in range(3)],
                       features={"normals": [G(inp["xn"][n, : xl[n]], dev) for n in range(3)]})
    pc_y = Pointclouds([G(inp["y"][n, : yl[n]], dev) for n in range(3)],
                       features={"normals": [G(inp["yn"][n, : yl[n]], dev) for n in range(3)]})
    loss, lf = chamfer_distance(pc_x, pc_y, feature_names=["normals"])
    assert close(loss.cpu().numpy(), g["pointclouds/loss"])
    assert close(lf["normals"].cpu().numpy(), g["pointclouds/lossf"])


def test_chamfer_zero_weights_and_errors(dev):
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

    x = G(cases.cloud(950, (2, 20, 3)), dev).requires_grad_(True)
    y = G(cases.cloud(951, (2, 30, 3)), dev)
    loss, lf = chamfer_distance(x, y, weights=torch.zeros(2, device=dev))
    assert float(loss) == 0.0 and lf is None
    loss.backward()
    assert float(x.grad.abs().sum()) == 0.0
    with pytest.raises(ValueError, match="cannot be negative"):
        chamfer_distance(x, y, weights=torch.tensor([1.0, -1.0], device=dev))
    with pytest.raises(ValueError, match="shape \\(N,\\)"):
        chamfer_distance(x, y, weights=torch.ones(3, device=dev))


# ------------------------------------------------------------------ full-size properties / digests
def test_cfg2_cloud_digest(dev):
    """One cfg2-size cloud (N=M=65536, K=16): sha256 of idx and dists equal the reference run."""
    from pytorch3d_pointops_amd.functions import knn_points

    meta = json.load(open(os.path.join(GOLDEN, "big_meta.json")))["cfg2_cloud"]
    P, K = meta["P"], meta["K"]
    p1 = cases.cloud(meta["seed1"], (1, P, 3))
    p2 = cases.cloud(meta["seed2"], (1, P, 3))
    r = knn_points(G(p1, dev), G(p2, dev), K=K)
    idx32 = r.idx.cpu().numpy().astype(np.int32)
    assert hashlib.sha256(idx32.tobytes()).hexdigest() == meta["idx_sha256"]
    assert hashlib.sha256(r.dists.cpu().numpy().tobytes()).hexdigest() == meta["dists_sha256"]


def test_cfg2_full_batch_properties(dev):
    """BASELINE.json configs[1] at full size (B=32, N=M=65536, K=16): size-independent properties.
    sortedness, index range, distances recomputed from idx bit-exactly, batch independence
    (cloud b of the batched call == the single-cloud call whose digest is pinned above)."""
    from pytorch3d_pointops_amd.functions import knn_points

    meta = json.load(open(os.path.join(GOLDEN, "big_meta.json")))["cfg2_cloud"]
    B, P, K = 32, meta["P"], meta["K"]
    p1 = np.empty((B, P, 3), np.float32)
    p2 = np.empty((B, P, 3), np.float32)
    for b in range(B):
        p1[b] = cases.cloud(meta["seed1"] + 10 * b, (P, 3))
        p2[b] = cases.cloud(meta["seed2"] + 10 * b, (P, 3))
    a, bq = G(p1, dev), G(p2, dev)
    r = knn_points(a, bq, K=K)
    d, idx = r.dists, r.idx
    assert bool((d[..., 1:] >= d[..., :-1]).all())
    assert bool(((idx >= 0) & (idx < P)).all())
    # ties ordered by index
    tie = d[..., 1:] == d[..., :-1]
    assert bool((idx[..., 1:][tie] > idx[..., :-1][tie]).all())
    # distances recomputed from the returned indices, unfused fp32, must be bit-equal
    nb = torch.gather(bq, 1, idx.reshape(B, P * K, 1).expand(-1, -1, 3)).reshape(B, P, K, 3)
    diff = a[:, :, None, :] - nb
    sq = diff * diff
    re = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    assert torch.equal(re, d)
    # cloud 0 equals the pinned single-cloud digest
    idx32 = idx[0:1].cpu().numpy().astype(np.int32)
    assert hashlib.sha256(idx32.tobytes()).hexdigest() == meta["idx_sha256"]
    # K-th distance is a true bound: count of points strictly closer than the K-th equals at most K-1
    # (checked on a sample of queries against a brute-force torch distance matrix)
    qs = torch.arange(0, P, 4099, device=dev)
    for b in (0, B - 1):
        dq = a[b, qs][:, None, :] - bq[b][None, :, :]
        dq = dq * dq
        full = (dq[..., 0] + dq[..., 1]) + dq[..., 2]
        kth = d[b, qs, K - 1]
        assert bool(((full < kth[:, None]).sum(1) <= K - 1).all())
        assert bool(((full <= kth[:, None]).sum(1) >= K).all())


def test_cfg3_fps_and_ball_query_full_cloud(dev):
    from pytorch3d_pointops_amd import _C

    meta = json.load(open(os.path.join(GOLDEN, "big_meta.json")))
    g = load_golden("big")
    m = meta["cfg3_fps"]
    pts = G(cases.cloud(m["seed"], (1, m["P"], 3)), dev)
    one = lambda v: torch.tensor([v], dtype=torch.int64, device=dev)
    idx = _C.sample_farthest_points(pts, one(m["P"]), one(m["K"]), one(0))
    assert np.array_equal(idx.cpu().numpy(), g["cfg3_fps/idx"].astype(np.int64))
    b = meta["cfg3_bq"]
    bi, bd = _C.ball_query(pts[:, : b["Q"]].contiguous(), pts, one(b["Q"]), one(b["P"]), b["K"], b["radius"])
    assert hashlib.sha256(bi.cpu().numpy().astype(np.int32).tobytes()).hexdigest() == b["idx_sha256"]
    assert hashlib.sha256(bd.cpu().numpy().tobytes()).hexdigest() == b["dists_sha256"]


def test_cfg3_ball_query_full_batch_properties(dev):
    """BASELINE.json configs[2] ball-query half at full size (B=16, N=131072, r=0.2, K=32):
    size-independent properties -- indices strictly ascending per row, every listed distance is the
    unfused fp32 distance of that pair and < r*r, -1 padding only after the hits, and on sampled
    queries the row is exactly the first K in-radius points of a brute-force torch scan."""
    from pytorch3d_pointops_amd import _C

    B, P, K, r = 16, 131072, 32, 0.2
    pts = np.empty((B, P, 3), np.float32)
    for b in range(B):
        pts[b] = cases.cloud(7003 + 10 * b, (P, 3))
    x = G(pts, dev)
    L = torch.full((B,), P, dtype=torch.int64, device=dev)
    idx, d = _C.ball_query(x, x, L, L, K, r)
    valid = idx >= 0
    r2 = np.float32(r) * np.float32(r)
    assert bool((d[valid] < float(r2)).all()) and bool((d[~valid] == 0).all())
    # hits first, then padding; indices ascending
    assert bool((valid[..., 1:] <= valid[..., :-1]).all())
    both = valid[..., 1:] & valid[..., :-1]
    assert bool((idx[..., 1:][both] > idx[..., :-1][both]).all())
    # distances recomputed from the indices (unfused fp32) are bit-equal
    safe = idx.clamp(min=0)
    nb = torch.gather(x, 1, safe.reshape(B, P * K, 1).expand(-1, -1, 3)).reshape(B, P, K, 3)
    diff = x[:, :, None, :] - nb
    sq = diff * diff
    re = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    assert torch.equal(re[valid], d[valid])
    # cloud 0, first 2048 queries == pinned reference digest (tests/golden/big_meta.json)
    meta = json.load(open(os.path.join(GOLDEN, "big_meta.json")))["cfg3_bq"]
    i0 = idx[0, : meta["Q"]].cpu().numpy().astype(np.int32)
    assert hashlib.sha256(i0.tobytes()).hexdigest() == meta["idx_sha256"]
    # sampled rows against a brute-force scan
    qs = torch.arange(5, P, 9973, device=dev)
    for b in (3, B - 1):
        dq = x[b, qs][:, None, :] - x[b][None, :, :]
        dq = dq * dq
        full = (dq[..., 0] + dq[..., 1]) + dq[..., 2]
        inside = full < float(r2)
        rank = torch.cumsum(inside.to(torch.int32), 1)
        for k in range(0, K, 7):
            want = torch.where((rank == k + 1) & inside, torch.arange(P, device=dev)[None], P).min(1).values
            want = torch.where(want == P, torch.full_like(want, -1), want)
            assert torch.equal(want, idx[b, qs, k])


def test_cfg4_chamfer_fused_vs_composed(dev, monkeypatch):
    """BASELINE.json configs[3] shape (B=8, ragged 20k..200k, normals, fwd+bwd): the fused chamfer
    direction (one forward + one backward kernel) against the composed path built from knn_points /
    knn_gather / torch ops that the small-size reference goldens pin -- values and all four gradients."""
    import pytorch3d_pointops_amd.functions.chamfer as ch
    from pytorch3d_pointops_amd import synth

    Bq = 8
    l1 = synth.randint(41, 20000, 200000, (Bq,))
    l2 = synth.randint(42, 20000, 200000, (Bq,))
    P1, P2 = int(l1.max()), int(l2.max())
    base = dict(x=synth.uniform_f32(43, (Bq, P1, 3)), y=synth.uniform_f32(44, (Bq, P2, 3)),
                xn=synth.unit_normals(45, (Bq, P1, 3)), yn=synth.unit_normals(46, (Bq, P2, 3)))
    w = G(np.linspace(0.5, 1.5, Bq).astype(np.float32), dev)

    def run():
        t = {k: G(v, dev).requires_grad_(True) for k, v in base.items()}
        loss, lf = ch.chamfer_distance(t["x"], t["y"], x_lengths=G(l1, dev), y_lengths=G(l2, dev),
                                       x_features={"normals": t["xn"]}, y_features={"normals": t["yn"]},
                                       feature_names=["normals"], weights=w, batch_reduction=None)
        (loss.sum() + lf["normals"].sum()).backward()
        return loss.detach(), lf["normals"].detach(), {k: v.grad for k, v in t.items()}

    a = run()
    monkeypatch.setattr(ch, "_fused_direction_ok", lambda *args, **kw: False)
    b = run()
    assert close(a[0].cpu().numpy(), b[0].cpu().numpy()) and close(a[1].cpu().numpy(), b[1].cpu().numpy())
    for k in base:
        assert close(a[2][k].cpu().numpy(), b[2][k].cpu().numpy(), tol=2e-5), k


@pytest.mark.parametrize("batch_reduction", [None, "mean", "sum"])
@pytest.mark.parametrize("with_normals", [False, True])
def test_chamfer_pair_native_vs_composed(dev, monkeypatch, batch_reduction, with_normals):
    """The one-call bidirectional chamfer (pointops_chamfer_pair_forward / _backward: both searches, both fused
    reductions, the sum of the directions, the batch reduction and its backward in native code) against the composed
    path that the reference goldens pin (knn_points / knn_gather / torch ops), ragged lengths, a missing gradient
    (only the point loss is back-propagated in one of the runs): values and all gradients."""
    import pytorch3d_pointops_amd.functions.chamfer as ch

    N, P1, P2 = 5, 700, 900
    l1, l2 = np.array([700, 1, 350, 699, 20]), np.array([900, 450, 1, 33, 899])
    base = dict(x=cases.cloud(2101, (N, P1, 3)), y=cases.cloud(2102, (N, P2, 3)))
    if with_normals:
        base.update(xn=cases.cloud(2103, (N, P1, 3)) - np.float32(0.5), yn=cases.cloud(2104, (N, P2, 3)) - np.float32(0.5))

    def run(point_reduction, only_points):
        t = {k: G(v, dev).requires_grad_(True) for k, v in base.items()}
        kw = dict(x_features={"normals": t["xn"]}, y_features={"normals": t["yn"]}, feature_names=["normals"]) \
            if with_normals else {}
        loss, lf = ch.chamfer_distance(t["x"], t["y"], x_lengths=G(l1, dev), y_lengths=G(l2, dev),
                                       batch_reduction=batch_reduction, point_reduction=point_reduction, **kw)
        total = loss.sum() * 1.5
        if with_normals and not only_points:
            total = total + lf["normals"].sum() * 0.25
        total.backward()
        return [loss.detach()] + ([lf["normals"].detach()] if with_normals else []), {k: v.grad for k, v in t.items()}

    for point_reduction in ("mean", "sum"):
        for only_points in (False, True):
            a = run(point_reduction, only_points)
            with monkeypatch.context() as m:
                m.setattr(ch, "_fused_direction_ok", lambda *args, **kw: False)
                b = run(point_reduction, only_points)
            for u, v in zip(a[0], b[0]):
                assert u.shape == v.shape and close(u.cpu().numpy(), v.cpu().numpy())
            for k in base:
                if a[1][k] is None or b[1][k] is None:  # (features without a back-propagated loss)
                    assert only_points and k in ("xn", "yn")
                    ga = a[1][k].cpu().numpy() if a[1][k] is not None else 0.0
                    gb = b[1][k].cpu().numpy() if b[1][k] is not None else 0.0
                    assert np.all(ga == 0.0) and np.all(gb == 0.0)
                    continue
                assert close(a[1][k].cpu().numpy(), b[1][k].cpu().numpy(), tol=2e-5), (k, point_reduction, only_points)


# ------------------------------------------------------------------ sample_pdf
@pytest.mark.parametrize("name", sorted(cases.sample_pdf_cases()))
def test_sample_pdf(dev, oracle, name):
    from pytorch3d_pointops_amd import _C

    g = load_golden("sample_pdf")
    c = cases.sample_pdf_cases()[name]
    out = G(c["u"].copy(), dev)
    _C.sample_pdf(G(c["bins"], dev), G(c["weights"], dev), out, c["eps"])
    got = out.cpu().numpy()
    assert np.array_equal(bits(got), bits(g[name + "/samples"]))
    assert np.array_equal(bits(got), bits(oracle.sample_pdf(c["bins"], c["weights"], c["u"], c["eps"])))


def test_sample_pdf_wrapper(dev):
    from pytorch3d_pointops_amd.functions.sample_pdf import sample_pdf, sample_pdf_python

    g = load_golden("sample_pdf")
    c = cases.sample_pdf_cases()["b8_64x128"]
    bins = G(c["bins"], dev).reshape(2, 4, -1)
    w = G(c["weights"], dev).reshape(2, 4, -1)
    det = sample_pdf(bins, w, 40, det=True)
    assert tuple(det.shape) == (2, 4, 40)
    assert np.array_equal(bits(det.cpu().numpy()), bits(g["wrapper_det/samples"]))
    # random quantiles: samples stay inside the bin range and roughly follow the python variant
    torch.manual_seed(0)
    rnd = sample_pdf(bins, w, 64)
    assert bool((rnd >= bins[..., :1]).all()) and bool((rnd <= bins[..., -1:]).all())
    assert close(sample_pdf_python(bins, w, 40, det=True).cpu().numpy(), det.cpu().numpy(), tol=1e-3)
    with pytest.raises(ValueError, match="Negative weights"):
        sample_pdf(bins, w - 1.0, 4)
    with pytest.raises(ValueError, match="Inconsistent shapes"):
        sample_pdf(bins[..., :-1], w, 4)
    with pytest.raises(NotImplementedError):
        sample_pdf(bins, w.clone().requires_grad_(True), 4)


# ------------------------------------------------------------------ round 2: launch shapes that had no parity check
def test_cfg3_fps_full_batch(dev, monkeypatch):
    """BASELINE.json configs[2] FPS half at full size (B=16, N=131072 -> 1024): the batched launch (16 clouds x
    16 workgroups = every CU, XCD-local clusters, per-iteration slot exchange) gives, for EVERY cloud, the
    indices of that cloud sampled alone; cloud 0 is the cloud whose reference output is pinned in
    tests/golden/big.npz.  Repeated for every cluster placement mode and with the exchange forced to time
    out (fps_spin_limit=0: all clouds are then redone by the single-workgroup repair pass)."""
    from pytorch3d_pointops_amd import _C

    meta = json.load(open(os.path.join(GOLDEN, "big_meta.json")))["cfg3_fps"]
    g = load_golden("big")
    B, P, K = 16, meta["P"], meta["K"]
    pts = np.empty((B, P, 3), np.float32)
    for b in range(B):
        pts[b] = cases.cloud(meta["seed"] + 10 * b, (P, 3))
    x = G(pts, dev)
    full = lambda n, v: torch.full((n,), v, dtype=torch.int64, device=dev)
    alone = torch.cat([_C.sample_farthest_points(x[b:b + 1], full(1, P), full(1, K), full(1, 0)) for b in range(B)])
    assert np.array_equal(alone[0].cpu().numpy(), g["cfg3_fps/idx"][0].astype(np.int64))
    for knobs in ("fps_mode=2", "fps_mode=1", "fps_mode=0", "fps_spin_limit=0"):
        monkeypatch.setenv("POINTOPS_DEBUG", knobs)
        batch = _C.sample_farthest_points(x, full(B, P), full(B, K), full(B, 0))
        assert torch.equal(batch, alone), knobs
    # ragged batch through the same launch: per-cloud lengths and K, start indices != 0
    monkeypatch.delenv("POINTOPS_DEBUG")
    lens = torch.tensor([P, 70000, 1, 131071] * 4, dtype=torch.int64, device=dev)
    ks = torch.tensor([K, 5, 3, 700] * 4, dtype=torch.int64, device=dev)
    st = torch.tensor([0, 69999, 0, 12345] * 4, dtype=torch.int64, device=dev)
    batch = _C.sample_farthest_points(x, lens, ks, st)
    for b in (1, 2, 3, 7):
        one = _C.sample_farthest_points(x[b:b + 1], lens[b:b + 1], ks[b:b + 1], st[b:b + 1])
        assert torch.equal(batch[b, : one.shape[1]], one[0]) and bool((batch[b, one.shape[1]:] == -1).all())


def test_cfg4_ragged_knn_vs_bruteforce_sample(dev):
    """BASELINE.json configs[3] shape (B=8, ragged 20k..200k): both K=1 searches of the chamfer at full size,
    sampled queries of every cloud against a brute-force torch distance row over the cloud's valid points
    (nearest index with ties to the lowest index, bit-equal distance); padded rows are zero."""
    from pytorch3d_pointops_amd import synth
    from pytorch3d_pointops_amd.functions import knn_points

    Bq = 8
    l1 = synth.randint(41, 20000, 200000, (Bq,))
    l2 = synth.randint(42, 20000, 200000, (Bq,))
    P1, P2 = int(l1.max()), int(l2.max())
    x, y = G(synth.uniform_f32(43, (Bq, P1, 3)), dev), G(synth.uniform_f32(44, (Bq, P2, 3)), dev)
    for a, b, la, lb in ((x, y, l1, l2), (y, x, l2, l1)):
        r = knn_points(a, b, G(la, dev), G(lb, dev), K=1)
        for n in range(Bq):
            qs = torch.arange(3, int(la[n]), 997, device=dev)
            dq = a[n, qs][:, None, :] - b[n, : int(lb[n])][None, :, :]
            dq = dq * dq
            full = (dq[..., 0] + dq[..., 1]) + dq[..., 2]
            best = full.min(1)
            first = torch.where(full == best.values[:, None], torch.arange(full.shape[1], device=dev)[None],
                                full.shape[1]).min(1).values
            assert torch.equal(r.idx[n, qs, 0], first) and torch.equal(r.dists[n, qs, 0], best.values), n
            assert bool((r.idx[n, int(la[n]):] == 0).all()) and bool((r.dists[n, int(la[n]):] == 0).all())


def test_cfg5_rank_workload_sharded_chamfer_nccl(dev):
    """BASELINE.json configs[4], one rank's share (32 clouds x 65536 points, fwd + bwd) through
    `sharded_chamfer_distance` on the `nccl` (RCCL) backend with world_size 1 -- the all_gather, its
    backward and the batch reduction on the real backend -- against chamfer_distance(batch_reduction="mean")
    on the same clouds: loss equal, gradients equal up to atomic order."""
    import torch.distributed as dist

    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance
    from pytorch3d_pointops_amd.sharded import sharded_chamfer_distance

    B, P = 32, 65536
    p1 = np.empty((B, P, 3), np.float32)
    p2 = np.empty((B, P, 3), np.float32)
    for b in range(B):
        p1[b] = cases.cloud(7001 + 10 * b, (P, 3))
        p2[b] = cases.cloud(7002 + 10 * b, (P, 3))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        xa, ya = G(p1, dev).requires_grad_(True), G(p2, dev).requires_grad_(True)
        xb, yb = G(p1, dev).requires_grad_(True), G(p2, dev).requires_grad_(True)
        ls, _ = sharded_chamfer_distance(xa, ya, B)
        ls.backward()
        lr, _ = chamfer_distance(xb, yb, batch_reduction="mean")
        lr.backward()
        assert abs(float(ls) - float(lr)) <= 1e-6 * max(1.0, abs(float(lr)))
        assert close(xa.grad.cpu().numpy(), xb.grad.cpu().numpy()) and close(ya.grad.cpu().numpy(), yb.grad.cpu().numpy())
        per, _ = sharded_chamfer_distance(xa.detach(), ya.detach(), B, batch_reduction=None)
        assert per.shape == (B,) and abs(float(per.mean()) - float(lr)) <= 1e-6
    finally:
        if started:
            dist.destroy_process_group()


def test_reference_example_call_patterns(dev):
    """The reference's own example script (examples/knn_on_pointclouds.py) at its sizes: Pointclouds batch of
    1500 + 800 points, self-KNN K=10 with per-cloud lengths, 200 x 800 cross query K=5, knn_gather of normals /
    colours and the inverse-distance interpolation -- against what the reference produced on these inputs
    (tests/golden/examples.npz, inputs stored in the fixture)."""
    from pytorch3d_pointops_amd.functions import knn_gather, knn_points
    from pytorch3d_pointops_amd.structures import Pointclouds

    g = load_golden("examples")
    pts = [G(g[f"in/points{i}"], dev) for i in range(2)]
    nrm = [G(g[f"in/normals{i}"], dev) for i in range(2)]
    col = [G(g[f"in/colors{i}"], dev) for i in range(2)]
    pc = Pointclouds(points=pts, features={"normals": nrm, "colors": col})
    padded, lens = pc.points_padded(), pc.num_points_per_cloud()
    assert np.array_equal(padded.cpu().numpy(), g["self/padded"])
    r = knn_points(padded, padded, lengths1=lens, lengths2=lens, K=10, return_nn=True)
    assert np.array_equal(r.idx.cpu().numpy(), g["self/idx"].astype(np.int64))
    assert np.array_equal(bits(r.dists.cpu().numpy()), bits(g["self/dists"]))
    assert np.array_equal(bits(r.knn.cpu().numpy()), bits(g["self/knn"]))
    c = knn_points(p1=pts[0][:200][None], p2=pts[1][None], K=5, return_nn=False)
    assert np.array_equal(c.idx.cpu().numpy(), g["cross/idx"].astype(np.int64))
    assert np.array_equal(bits(c.dists.cpu().numpy()), bits(g["cross/dists"]))
    gn = knn_gather(nrm[1][None], c.idx)[0]
    gc = knn_gather(col[1][None], c.idx)[0]
    assert np.array_equal(bits(gn.cpu().numpy()), bits(g["cross/gathered_normals"]))
    assert np.array_equal(bits(gc.cpu().numpy()), bits(g["cross/gathered_colors"]))
    w = 1.0 / (torch.sqrt(c.dists[0]) + 1e-8)
    w = w / w.sum(dim=1, keepdim=True)
    interp_n = torch.nn.functional.normalize((gn * w.unsqueeze(-1)).sum(dim=1), p=2, dim=1)
    assert close(interp_n.cpu().numpy(), g["cross/interp_normals"]) and close(
        (gc * w.unsqueeze(-1)).sum(dim=1).cpu().numpy(), g["cross/interp_colors"])


def test_self_query_reuses_point_sort_cfg1_pattern(dev, oracle, monkeypatch):
    """cfg1 call pattern (knn_points(p, p): the same tensor twice) at a size the grid takes: identical to the
    p1 != p2 machinery and to the oracle; also through ball_query and get_point_covariances."""
    from pytorch3d_pointops_amd.functions import ball_query, knn_points

    p = cases.cloud(2101, (3, 9000, 3))
    p[1] = (p[1] ** np.float32(3.0)).astype(np.float32)
    lens = np.array([9000, 4321, 17])
    t, lt = G(p, dev), G(lens, dev)
    r = knn_points(t, t, lt, lt, K=16, version=3)
    oi, od = oracle.knn_points_idx(p, p, lens, lens, 2, 16)
    assert np.array_equal(r.idx.cpu().numpy(), oi) and np.array_equal(bits(r.dists.cpu().numpy()), bits(od))
    monkeypatch.setenv("POINTOPS_DEBUG", "ball_grid=1,ball_factor=0")
    b = ball_query(t, t, lt, lt, K=12, radius=0.05, return_nn=False)
    bi, bd = oracle.ball_query(p, p, lens, lens, 12, 0.05)
    assert np.array_equal(b.idx.cpu().numpy(), bi) and np.array_equal(bits(b.dists.cpu().numpy()), bits(bd))


def test_registered_ops_and_torch_compile(dev):
    """The operators are registered torch ops (torch.ops.pointops_amd.*) with fake implementations and autograd:
    `torch.compile(knn_points)` traces through them and returns what the eager call returns -- values and
    gradients; opcheck-style: the op list covers the `_C` operator boundary."""
    from pytorch3d_pointops_amd import ops
    from pytorch3d_pointops_amd.functions import knn_gather, knn_points

    need = {"knn_points_idx", "knn_points_backward", "ball_query", "sample_farthest_points", "packed_to_padded",
            "padded_to_packed", "gather_neighbors", "gather_neighbors_backward", "sample_pdf"}
    assert need <= set(ops.registered_ops())
    p1 = G(cases.cloud(2201, (2, 700, 3)), dev)
    p2 = G(cases.cloud(2202, (2, 900, 3)), dev)
    feats = G(cases.cloud(2203, (2, 900, 5)), dev)
    l1, l2 = G(np.array([700, 301]), dev), G(np.array([900, 555]), dev)

    def fn(a, b, f):
        r = knn_points(a, b, l1, l2, K=6, return_nn=True)
        return r.dists, r.idx, r.knn, knn_gather(f, r.idx, l2)

    a1, b1, f1 = (t.clone().requires_grad_(True) for t in (p1, p2, feats))
    e = fn(a1, b1, f1)
    (e[0].sum() + (e[2] * 0.5).sum() + e[3].sum()).backward()
    a2, b2, f2 = (t.clone().requires_grad_(True) for t in (p1, p2, feats))
    c = torch.compile(fn, fullgraph=True)(a2, b2, f2)
    (c[0].sum() + (c[2] * 0.5).sum() + c[3].sum()).backward()
    for u, v in zip(e, c):
        assert torch.equal(u, v)
    assert torch.equal(a1.grad, a2.grad) and close(b1.grad.cpu().numpy(), b2.grad.cpu().numpy())
    assert close(f1.grad.cpu().numpy(), f2.grad.cpu().numpy())
    # the raw op is differentiable on its own as well
    a3 = p1.clone().requires_grad_(True)
    idx, d = torch.ops.pointops_amd.knn_points_idx(a3, p2, l1, l2, 2, 6, -1)
    d.sum().backward()
    assert torch.equal(idx, e[1]) and torch.equal(a3.grad, torch.autograd.grad(fn(a1, b1, f1)[0].sum(), a1)[0])


def test_deterministic_backward_passes(dev, oracle):
    """torch.use_deterministic_algorithms(True): the scatter sides of the backward passes run through the inverted
    neighbour table (csrc/backward_det.hip) -- two runs are bit-identical, grad_p2 is BIT-EQUAL to the reference's CPU
    backward (the knn_backward.npz goldens, and the oracle on a table with hubs, -1 padding and ragged lengths), and
    chamfer_distance (composed path) and knn_gather train under the flag instead of raising."""
    from pytorch3d_pointops_amd import _C
    from pytorch3d_pointops_amd.functions import ball_query, knn_gather, knn_points
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

    g = load_golden("knn_backward")
    for name, c in sorted(cases.knn_backward_cases().items()):
        p1t, p2t, l1t, l2t = G(c["p1"], dev), G(c["p2"], dev), G(c["l1"], dev), G(c["l2"], dev)
        idx, dists = _C.knn_points_idx(p1t, p2t, l1t, l2t, c["norm"], c["K"], -1)
        grad = G(cases.grad_for(name, tuple(dists.shape)), dev)
        g1, g2 = _C.knn_points_backward(p1t, p2t, l1t, l2t, idx, c["norm"], grad, deterministic=True)
        assert np.array_equal(bits(g1.cpu().numpy()), bits(g[name + "/grad_p1"])), name
        assert np.array_equal(bits(g2.cpu().numpy()), bits(g[name + "/grad_p2"])), name  # (1e-5 with atomics)
    # hubs (every query's nearest neighbours are the same few points), ragged lengths, ball-query padding
    p1 = cases.cloud(3401, (3, 3000, 3))
    p2 = cases.cloud(3402, (3, 500, 3))
    p2[1, 5:] += np.float32(10.0)  # cloud 1: five hubs take every entry
    l1, l2 = np.array([3000, 2000, 7]), np.array([500, 500, 3])
    for norm in (2, 1):
        oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, 8)
        gd = cases.cloud(3403, (3, 3000, 8))
        e1, e2 = oracle.knn_points_backward(p1, p2, l1, l2, oi, norm, gd)
        a = _C.knn_points_backward(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), G(oi, dev), norm, G(gd, dev),
                                   deterministic=True)
        assert np.array_equal(bits(a[0].cpu().numpy()), bits(e1)) and np.array_equal(bits(a[1].cpu().numpy()), bits(e2))
    bi, bd = oracle.ball_query(p1, p2, l1, l2, 6, 0.3)
    gd = cases.cloud(3404, (3, 3000, 6))
    e1, e2 = oracle.knn_points_backward(p1, p2, l1, l2, bi, 2, gd)
    a = _C.knn_points_backward(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), G(bi, dev), 2, G(gd, dev),
                               deterministic=True)
    assert np.array_equal(bits(a[0].cpu().numpy()), bits(e1)) and np.array_equal(bits(a[1].cpu().numpy()), bits(e2))

    x = G(cases.cloud(2301, (2, 3000, 3)), dev)
    y = G(cases.cloud(2302, (2, 4000, 3)), dev)
    xn, yn = G(cases.cloud(2303, (2, 3000, 3)), dev), G(cases.cloud(2304, (2, 4000, 3)), dev)

    def train_step():
        t = [v.clone().requires_grad_(True) for v in (x, y, xn, yn)]
        r = knn_points(t[0], t[1], K=4)
        b = ball_query(t[0], t[1], K=5, radius=0.1, return_nn=False)
        loss, lf = chamfer_distance(t[0], t[1], x_features={"n": t[2]}, y_features={"n": t[3]}, feature_names=["n"])
        total = r.dists.sum() + b.dists.sum() + knn_gather(t[3], r.idx).square().sum() + loss + lf["n"]
        total.backward()
        return [v.grad.clone() for v in t]

    ref = train_step()  # default (atomics / LDS tiles)
    try:
        torch.use_deterministic_algorithms(True)
        d1, d2 = train_step(), train_step()
    finally:
        torch.use_deterministic_algorithms(False)
    for u, v, w in zip(d1, d2, ref):
        assert torch.equal(u, v)  # reproducible, bit for bit
        assert close(u.cpu().numpy(), w.cpu().numpy(), tol=2e-5)  # and the same gradients as the default passes
    # the fused chamfer kernels still announce their atomics when called directly under the flag
    from pytorch3d_pointops_amd.functions._common import alert_not_deterministic

    try:
        torch.use_deterministic_algorithms(True)
        with pytest.raises(RuntimeError, match="does not have a deterministic implementation"):
            alert_not_deterministic("chamfer_distance backward")
        torch.use_deterministic_algorithms(True, warn_only=True)
        with pytest.warns(UserWarning, match="does not have a deterministic implementation"):
            alert_not_deterministic("chamfer_distance backward")
    finally:
        torch.use_deterministic_algorithms(False)


def test_chamfer_device_and_layout_checks(dev):
    """pointers handed to the chamfer kernels are checked -- CPU weights / features raise instead of
    faulting, strided weights give the contiguous result."""
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

    x = G(cases.cloud(2401, (4, 200, 3)), dev)
    y = G(cases.cloud(2402, (4, 260, 3)), dev)
    xn, yn = G(cases.cloud(2403, (4, 200, 3)), dev), G(cases.cloud(2404, (4, 260, 3)), dev)
    w = torch.tensor([0.5, 1.0, 2.0, 0.25])
    with pytest.raises(RuntimeError):
        chamfer_distance(x, y, weights=w)  # CPU weights
    with pytest.raises(RuntimeError):
        chamfer_distance(x, y, x_features={"n": xn.cpu()}, y_features={"n": yn}, feature_names=["n"])
    wide = torch.stack([w, w * 7], dim=1).to(dev)  # (4, 2): column 0 is a strided view
    l_s, _ = chamfer_distance(x, y, weights=wide[:, 0])
    l_c, _ = chamfer_distance(x, y, weights=w.to(dev))
    assert torch.equal(l_s, l_c)


@pytest.mark.parametrize("long_box", ["auto", "0", "1"])
@pytest.mark.parametrize("K,norm", [(33, 2), (40, 2), (64, 2), (64, 1)])
def test_knn_grid_long_lists(dev, oracle, monkeypatch, K, norm, long_box):
    """K in (32, 64] through the grid family (64-slot lists in the lane search, no quad pass, the long-list
    brute-force kernel as the exact fallback for what stays uncertified): ragged clouds, a cluster, a cloud
    shorter than K and one without a usable grid, against the oracle and against the brute-force family.
    `grid_long_box` = 1 / 0 forces the two routes of the uncertified queries (box search -- what big batches take
    by themselves -- / wave search); with the box route the diagnostics must show queries deferred to it."""
    from pytorch3d_pointops_amd import _C
    from pytorch3d_pointops_amd.functions import knn_points

    if long_box != "auto":
        monkeypatch.setenv("POINTOPS_DEBUG", "grid_long_box=" + long_box)

    p1 = cases.cloud(2501, (4, 2500, 3))
    p2 = cases.cloud(2502, (4, 12000, 3))
    p2[1] = (p2[1] ** np.float32(3.0)).astype(np.float32)  # clustered
    p2[3, :, :] = p2[3, :1, :]  # all identical: one cell, everything ties
    l1 = np.array([2500, 2500, 900, 300])
    l2 = np.array([12000, 7000, K - 3, 12000])
    r = knn_points(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm=norm, K=K, version=3)
    oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
    assert np.array_equal(r.idx.cpu().numpy(), oi)
    assert np.array_equal(bits(r.dists.cpu().numpy()), bits(od))
    r0 = knn_points(G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm=norm, K=K, version=0)
    assert torch.equal(r0.idx, r.idx) and torch.equal(r0.dists, r.dists)
    if long_box == "1":
        # column 8 = queries handed to the box search, column 5 = the wave search's list (the lane pass's uncertified
        # queries without the box route, what the box search could not certify with it)
        args = (G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm, K)
        box_on = _C.knn_grid_stats(*args)[2].cpu().numpy()
        monkeypatch.setenv("POINTOPS_DEBUG", "grid_long_box=0")
        box_off = _C.knn_grid_stats(*args)[2].cpu().numpy()
        assert int(box_off[:2, 5].sum()) > 0, box_off  # there ARE uncertified queries on these clouds
        # (the wave list of the run without the route also holds the few queries its box search gave up on)
        assert int(box_on[:2, 8].sum()) > int(box_off[:2, 8].sum()), (box_on, box_off)
        assert int(box_on[:2, 8].sum()) >= int(box_off[:2, 5].sum()) - int(box_off[:2, 8].sum()), (box_on, box_off)


@pytest.mark.parametrize("K,norm", [(1, 2), (8, 2), (16, 1), (40, 2)])
def test_knn_refined_cells_and_box_search(dev, oracle, monkeypatch, K, norm):
    """Clouds whose density varies by orders of magnitude (half of the points in a 1e-3 cube; u^5 per coordinate):
    over-full cells are refined into sub-grids and the queries of over-full neighbourhoods go through the box
    search (grid_refine.hip, knn_grid_box.h).  The diagnostics confirm that path ran; results equal the oracle,
    the run without refinement (boxes over whole cells) and the brute-force family, bit for bit."""
    from pytorch3d_pointops_amd import _C

    m = 24000
    a = cases.cloud(2601, (3, 5000, 3))
    b = cases.cloud(2602, (3, m, 3))
    a[0, :2500] = a[0, :2500] * np.float32(1e-3) + np.float32(0.5)
    b[0, : m // 2] = b[0, : m // 2] * np.float32(1e-3) + np.float32(0.5)
    a[1], b[1] = (a[1] ** np.float32(5.0)).astype(np.float32), (b[1] ** np.float32(5.0)).astype(np.float32)
    b[2, : m // 3] = b[2, 0]  # a third of the cloud is ONE point: every sub-cell index ties
    l1, l2 = np.array([5000, 5000, 1234]), np.array([m, m, m - 5])
    args = (G(a, dev), G(b, dev), G(l1, dev), G(l2, dev), norm, K)
    idx, d, st = _C.knn_grid_stats(*args)
    st = st.cpu().numpy()
    assert (st[:2, 8] > 0).all() and (st[:2, 9] > 0).all(), st  # deferred to the box search, refined cells
    oi, od = oracle.knn_points_idx(a, b, l1, l2, norm, K)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))
    monkeypatch.setenv("POINTOPS_DEBUG", "grid_refine=0")
    idx0, d0, st0 = _C.knn_grid_stats(*args)
    assert int(st0[:, 9].sum()) == 0 and torch.equal(idx0, idx) and torch.equal(d0, d)
    monkeypatch.delenv("POINTOPS_DEBUG")
    i2, d2 = _C.knn_points_idx(*args, 2 if K <= 32 else 0)
    assert torch.equal(i2, idx) and torch.equal(d2, d)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [3, 2, 1])
def test_knn_grid_many_crowded_bins(dev, D):
    """The two-level sort of the grid build (grid_build.hip) on a cloud with MORE crowded bins than it lists (100 tight
    clusters of 9 000 points: every cluster is a bin of its own, 64 of them are placed by slices with the per-cell ranks
    of the scatter launch, the rest by one workgroup each), as points and as queries (self-query included): sampled
    queries against a brute-force torch distance row over the whole cloud, nearest neighbours with ties to the lowest
    index, bit-equal distances."""
    from pytorch3d_pointops_amd.functions import knn_points

    rng = np.random.default_rng(2701)
    centres = rng.random((100, D), dtype=np.float32)
    pts = (centres[:, None, :] + rng.random((100, 9000, D), dtype=np.float32) * np.float32(2e-4)).reshape(-1, D)
    pts = np.concatenate([pts, rng.random((50000, D), dtype=np.float32)])
    rng.shuffle(pts)
    p2 = G(pts[None], dev)
    q = np.concatenate([pts[::37][:20000] + np.float32(1e-5), rng.random((10000, D), dtype=np.float32),
                        centres[:1] + rng.random((10000, D), dtype=np.float32) * np.float32(2e-4)])  # a crowded query bin
    p1 = G(q[None].astype(np.float32), dev)
    K = 4
    from pytorch3d_pointops_amd import _C
    full_len = lambda t: torch.full((1,), t.shape[1], dtype=torch.int64, device=dev)  # noqa: E731
    st = _C.knn_grid_stats(p1, p2, full_len(p1), full_len(p2), 2, K)[2].cpu().numpy()[0]
    assert st[4] == 1 and st[10] >= 100 and st[12] == 64 and st[13] >= 1, st  # > 64 crowded bins: the list is full
    for a in (p1, p2):  # p1 != p2, and the self-query (the point sort is the query order)
        r = knn_points(a, p2, K=K, version=3)
        qs = torch.arange(5, a.shape[1], max(1, a.shape[1] // 300), device=dev)
        dq = a[0, qs][:, None, :] - p2[0][None, :, :]
        dq = dq * dq
        full = dq[..., 0]
        if D >= 2:
            full = full + dq[..., 1]
        if D == 3:
            full = full + dq[..., 2]
        order = torch.argsort(full, dim=1, stable=True)[:, :K]
        assert torch.equal(r.idx[0, qs], order), D
        assert torch.equal(r.dists[0, qs], torch.gather(full, 1, order)), D


@pytest.mark.gpu
def test_chamfer_backward_accumulate_equals_sum_of_directions(dev):
    """pointops_chamfer_backward_accumulate (include/pointops_amd.h): the reverse direction's gradients ADDED into the
    buffers the forward direction's call filled equal the sum of two separate backward calls (the reference sums the
    directions through autograd, functions/chamfer.py:316-354) -- dense terms exactly up to one fp32 addition, the
    atomically scattered ones within the 1e-5 of every atomic accumulation; padded rows stay zero."""
    from pytorch3d_pointops_amd import _C

    N, P1, P2 = 3, 700, 900
    x, y = G(cases.cloud(3101, (N, P1, 3)), dev), G(cases.cloud(3102, (N, P2, 3)), dev)
    xf = [G(cases.cloud(3103, (N, P1, 3)), dev)]
    yf = [G(cases.cloud(3104, (N, P2, 3)), dev)]
    xl, yl = G(np.array([700, 333, 1]), dev), G(np.array([900, 10, 450]), dev)
    g = G(cases.cloud(3105, (2, N)), dev)
    idx_xy = _C.knn_points_idx(x, y, xl, yl, 2, 1, -1)[0][..., 0].contiguous()
    idx_yx = _C.knn_points_idx(y, x, yl, xl, 2, 1, -1)[0][..., 0].contiguous()
    gx, gy, gxf, gyf = _C.chamfer_backward(x, y, idx_xy, xl, yl, None, g, 2, xf, yf, True, True)
    gy2, gx2, gyf2, gxf2 = _C.chamfer_backward(y, x, idx_yx, yl, xl, None, g, 2, yf, xf, True, True)
    want = [gx + gx2, gy + gy2, gxf[0] + gxf2[0], gyf[0] + gyf2[0]]
    _C.chamfer_backward(y, x, idx_yx, yl, xl, None, g, 2, yf, xf, True, True, into=(gy, gx, gyf, gxf))
    for got, ref in zip([gx, gy, gxf[0], gyf[0]], want):
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)
    assert bool((gx[1, 333:] == 0).all()) and bool((gy[1, 10:] == 0).all())
    with pytest.raises(RuntimeError, match="into"):
        _C.chamfer_backward(y, x, idx_yx, yl, xl, None, g, 2, yf, xf, True, True, into=(gx, gy, gxf, gyf))


def test_integer_arguments_must_be_int64(dev):
    """The kernels read lengths / idx / K / start_idxs as int64 through raw pointers; an int32 tensor would be read
    past its buffer and give wrong neighbours with rc 0.  Every operator raises instead, like the reference's
    accessor<int64_t, 1> (knn_cpu.cpp:84-88, ball_query_cpu.cpp:28-29, sample_farthest_points_cpu.cpp:33-36)."""
    from pytorch3d_pointops_amd import _C
    from pytorch3d_pointops_amd.functions import ball_query

    p = G(cases.cloud(3301, (2, 64, 3)), dev)
    L = torch.full((2,), 64, dtype=torch.int64, device=dev)
    L32 = L.to(torch.int32)
    idx, d = _C.knn_points_idx(p, p, L, L, 2, 4, -1)
    g = torch.ones_like(d)
    with pytest.raises(RuntimeError, match="int64"):
        _C.knn_points_idx(p, p, L32, L, 2, 4, -1)
    with pytest.raises(RuntimeError, match="int64"):
        _C.ball_query(p, p, L32, L, 4, 0.2)
    with pytest.raises(RuntimeError, match="int64"):
        _C.ball_query(p, p, L, L32, 4, 0.2)
    with pytest.raises(RuntimeError, match="int64"):
        ball_query(p, p, L32, L32, K=4, radius=0.2)
    with pytest.raises(RuntimeError, match="int64"):
        _C.knn_points_backward(p, p, L32, L, idx, 2, g)
    with pytest.raises(RuntimeError, match="int64"):
        _C.knn_points_backward(p, p, L, L, idx.to(torch.int32), 2, g)
    K = torch.full((2,), 8, dtype=torch.int64, device=dev)
    S = torch.zeros((2,), dtype=torch.int64, device=dev)
    for bad in ((L32, K, S), (L, K.to(torch.int32), S), (L, K, S.to(torch.int32))):
        with pytest.raises(RuntimeError, match="int64"):
            _C.sample_farthest_points(p, *bad)
    with pytest.raises(RuntimeError, match="inconsistent shapes"):
        _C.ball_query(p, p[:1], L, L, 4, 0.2)
    assert _C.sample_farthest_points(p, L, K, S).shape == (2, 8)


def test_cfg4_smallest_pair_loss_and_gradients_vs_oracle(dev, oracle):
    """BASELINE.json configs[3] at full size (B=8 ragged 20k..200k, normals, fwd+bwd): the per-cloud loss
    (batch_reduction=None) and all four gradients of the batch's SMALLEST cloud pair against a chamfer built from the
    CPU oracle alone (both K=1 searches + knn_points_backward; the cosine term and its gradient in numpy), 1e-5."""
    from pytorch3d_pointops_amd import synth
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

    Bq = 8
    l1 = synth.randint(41, 20000, 200000, (Bq,))
    l2 = synth.randint(42, 20000, 200000, (Bq,))
    P1, P2 = int(l1.max()), int(l2.max())
    base = dict(x=synth.uniform_f32(43, (Bq, P1, 3)), y=synth.uniform_f32(44, (Bq, P2, 3)),
                xn=synth.unit_normals(45, (Bq, P1, 3)), yn=synth.unit_normals(46, (Bq, P2, 3)))
    wv = np.linspace(0.5, 1.5, Bq).astype(np.float32)
    t = {k: G(v, dev).requires_grad_(True) for k, v in base.items()}
    loss, lf = chamfer_distance(t["x"], t["y"], x_lengths=G(l1, dev), y_lengths=G(l2, dev),
                                x_features={"normals": t["xn"]}, y_features={"normals": t["yn"]},
                                feature_names=["normals"], weights=G(wv, dev), batch_reduction=None)
    (loss.sum() + lf["normals"].sum()).backward()

    n = int(np.argmin(l1.astype(np.int64) * l2.astype(np.int64)))
    a, b = int(l1[n]), int(l2[n])
    x, y = base["x"][n:n + 1, :a], base["y"][n:n + 1, :b]
    xn, yn = base["xn"][n, :a].astype(np.float64), base["yn"][n, :b].astype(np.float64)
    la, lb = np.array([a]), np.array([b])
    i1, d1 = oracle.knn_points_idx(x, y, la, lb, 2, 1)
    i2, d2 = oracle.knn_points_idx(y, x, lb, la, 2, 1)
    w = float(wv[n])
    exp_loss = w * (d1.astype(np.float64).sum() / a + d2.astype(np.float64).sum() / b)
    assert abs(float(loss[n]) - exp_loss) <= 1e-5 * max(1.0, abs(exp_loss))

    def cos_term(u, v):  # 1 - |cos(u, v)| per row and its gradients (torch: dot / max(|u| |v|, eps), eps = 1e-6)
        nu, nv = np.linalg.norm(u, axis=1), np.linalg.norm(v, axis=1)
        den = np.maximum(nu * nv, 1e-6)
        c = (u * v).sum(1) / den
        sgn = np.sign(c)
        gu = -(sgn / den)[:, None] * (v - (c * nv / nu)[:, None] * u)
        gv = -(sgn / den)[:, None] * (u - (c * nu / nv)[:, None] * v)
        return 1.0 - np.abs(c), gu, gv

    j1, j2 = i1[0, :, 0], i2[0, :, 0]
    tx, gxa, gyb = cos_term(xn, yn[j1])
    ty, gyc, gxd = cos_term(yn, xn[j2])
    exp_feat = w * (tx.sum() / a + ty.sum() / b)
    assert abs(float(lf["normals"][n]) - exp_feat) <= 1e-5 * max(1.0, abs(exp_feat))
    # point gradients: knn backward of both directions with grad_dists = w / length
    g1 = np.full(d1.shape, w / a, np.float32)
    g2 = np.full(d2.shape, w / b, np.float32)
    ax, ay = oracle.knn_points_backward(x, y, la, lb, i1, 2, g1)
    by, bx = oracle.knn_points_backward(y, x, lb, la, i2, 2, g2)
    assert close(t["x"].grad[n, :a].cpu().numpy(), (ax + bx)[0], tol=1e-5)
    assert close(t["y"].grad[n, :b].cpu().numpy(), (ay + by)[0], tol=1e-5)
    assert not bool(t["x"].grad[n, a:].any()) and not bool(t["y"].grad[n, b:].any())
    # normal gradients: dense rows of the own direction + the scatter of the other one
    gxn = (w / a) * gxa
    np.add.at(gxn, j2, (w / b) * gxd)
    gyn = (w / b) * gyc
    np.add.at(gyn, j1, (w / a) * gyb)
    assert close(t["xn"].grad[n, :a].cpu().numpy(), gxn.astype(np.float32), tol=1e-5)
    assert close(t["yn"].grad[n, :b].cpu().numpy(), gyn.astype(np.float32), tol=1e-5)


def test_reference_example_call_patterns_2(dev):
    """The call patterns of five more of the reference's self-checking example scripts at their own sizes
    (ball_query_on_pointclouds.py:50-125, fps_on_pointclouds.py:66-215, chamfer_loss.py:13-89,
    packed_to_padded_on_pointclouds.py:67-124, utils_on_pointclouds.py:69-237) against the outputs the REFERENCE
    produced for the same inputs (tests/golden/examples2.npz + examples2_meta.json, made by make_golden.py
    gen_examples2 in the build container): indices, distances, gathers and copies bit-exact, losses / covariances /
    weighted means within 1e-5."""
    import hashlib
    import json
    import os

    from conftest import GOLDEN
    from pytorch3d_pointops_amd.functions import (ball_query, knn_points, masked_gather, packed_to_padded,
                                                  padded_to_packed, sample_farthest_points)
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance
    from pytorch3d_pointops_amd.functions.sample_farthest_points import sample_farthest_points_naive
    from pytorch3d_pointops_amd.functions.utils import get_point_covariances, wmean
    from pytorch3d_pointops_amd.structures import Pointclouds

    g = load_golden("examples2")
    meta = json.load(open(os.path.join(GOLDEN, "examples2_meta.json")))
    inp = cases.example2_inputs()

    def same(tag, t):
        a = t.detach().cpu().numpy()
        if a.dtype == np.int64:
            a = a.astype(np.int32)
        if tag in meta:
            assert list(a.shape) == meta[tag]["shape"] and str(a.dtype) == meta[tag]["dtype"], tag
            flat = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a
            assert np.array_equal(np.ascontiguousarray(flat[::37]).view(np.int32), g[tag + "/rows"].view(np.int32)), tag
            assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == meta[tag]["sha256"], tag
        else:
            assert a.shape == g[tag].shape and np.array_equal(a.view(np.int32), g[tag].view(np.int32)), tag

    # ---- ball query
    b = inp["ball"]
    pc = Pointclouds(points=[G(a, dev) for a in b["points"]])
    pad, lens = pc.points_padded(), pc.num_points_per_cloud()
    r = ball_query(p1=pad, p2=pad, lengths1=lens, lengths2=lens, K=50, radius=0.3, return_nn=True)
    same("ball/dists", r.dists), same("ball/idx", r.idx), same("ball/knn", r.knn)
    valid = r.idx[0] != -1
    assert bool((torch.sqrt(r.dists[0][valid]) <= 0.3).all())  # the script's own check (:113-117)
    lat = G(b["lattice"], dev)[None]
    r = ball_query(p1=lat, p2=lat, K=30, radius=0.25, return_nn=False)
    k = knn_points(p1=lat, p2=lat, K=10, return_nn=False)
    same("lattice/ball_dists", r.dists), same("lattice/ball_idx", r.idx)
    same("lattice/knn_dists", k.dists), same("lattice/knn_idx", k.idx)
    # ---- FPS (random starts: the same torch.manual_seed as the generator, the same RNG consumption as the reference)
    f = inp["fps"]
    torch.manual_seed(123)
    sp, si = sample_farthest_points(G(f["single"], dev)[None], K=50, random_start_point=True)
    same("fps/single_idx", si), same("fps/single_points", sp)
    pc = Pointclouds(points=[G(a, dev) for a in f["points"]])
    torch.manual_seed(42)
    sp, si = sample_farthest_points(pc.points_padded(), lengths=pc.num_points_per_cloud(), K=[100, 80, 150],
                                    random_start_point=True)
    same("fps/batch_idx", si), same("fps/batch_points", sp)
    so, io = sample_farthest_points(G(f["compare"], dev), K=200, random_start_point=False)
    sn, in_ = sample_farthest_points_naive(G(f["compare"], dev), K=200, random_start_point=False)
    assert torch.equal(io, in_) and torch.equal(so, sn)  # the script's own check (:141-149)
    same("fps/compare_idx", io)
    torch.manual_seed(7)
    sp, si = sample_farthest_points(G(f["points"][0], dev)[None], K=100, random_start_point=True)
    same("fps/colors_idx", si), same("fps/colors", masked_gather(G(f["colors0"], dev)[None], si))
    sn, in_ = sample_farthest_points_naive(G(f["circles"], dev)[None], K=50, random_start_point=False)
    same("fps/circles_idx", in_)
    _, ih = sample_farthest_points(G(f["circles"], dev)[None], K=50)  # (the HIP kernel on the 2-D cloud)
    assert torch.equal(ih, in_)
    # ---- chamfer
    c = {kk: G(v, dev) for kk, v in inp["chamfer"].items()}
    l, lf = chamfer_distance(c["p1"], c["p2"], x_features={"normals": c["n1"], "colors": c["c1"]},
                             y_features={"normals": c["n2"], "colors": c["c2"]}, feature_names=["normals", "colors"])
    assert close(np.array([float(l), float(lf["normals"]), float(lf["colors"])]), g["chamfer/tensor"], tol=1e-5)
    pc1 = Pointclouds(points=list(c["p1"]), features={"normals": list(c["n1"]), "colors": list(c["c1"])})
    pc2 = Pointclouds(points=list(c["p2"]), features={"normals": list(c["n2"]), "colors": list(c["c2"])})
    l, lf = chamfer_distance(pc1, pc2, feature_names=["normals", "colors"])
    assert close(np.array([float(l), float(lf["normals"]), float(lf["colors"])]), g["chamfer/pointclouds"], tol=1e-5)
    l, lf = chamfer_distance(pc1, pc2, feature_names=["normals"], single_directional=True)
    assert close(np.array([float(l), float(lf["normals"])]), g["chamfer/single"], tol=1e-5)
    # ---- packed <-> padded
    kq = inp["packed"]
    pc = Pointclouds(points=[G(a, dev) for a in kq["points"]],
                     features={"intensities": [G(a, dev) for a in kq["intensities"]]})
    lens = pc.num_points_per_cloud()
    first = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), lens.cumsum(0)[:-1]])
    pp, ip = pc.points_packed(), pc.get_features_packed("intensities")
    pad, ipad = packed_to_padded(pp, first, int(lens.max())), packed_to_padded(ip, first, int(lens.max()))
    same("packed/points_packed", pp), same("packed/points_padded", pad), same("packed/intensities_padded", ipad)
    same("packed/points_repacked", padded_to_packed(pad, first, int(lens.sum())))
    same("packed/intensities_repacked", padded_to_packed(ipad, first, int(lens.sum())))
    pc = Pointclouds(points=[G(a, dev) for a in kq["var_points"]],
                     features={"features": [G(a, dev) for a in kq["var_features"]]})
    lens = pc.num_points_per_cloud()
    first = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), lens.cumsum(0)[:-1]])
    fpad = packed_to_padded(pc.get_features_packed("features"), first, int(lens.max()))
    same("packed/var_features_padded", fpad)
    same("packed/var_features_repacked", padded_to_packed(fpad, first, int(lens.sum())))
    # ---- utils
    u = inp["utils"]
    pc = Pointclouds(points=[G(a, dev) for a in u["points"]])
    cov, nn = get_point_covariances(pc.points_padded(), pc.num_points_per_cloud(), 16)
    assert close(cov.cpu().numpy()[:, ::5], g["utils/cov"], tol=1e-5)
    same("utils/cov_knn", nn)
    wm = np.stack([wmean(G(p, dev), G(w, dev).squeeze(), dim=0, keepdim=False).cpu().numpy()
                   for p, w in zip(u["points"], u["weights"])])
    assert close(wm, g["utils/wmean"], tol=1e-5)
    p0 = G(u["points"][0], dev)[None]
    kn = knn_points(p0, p0, K=8, return_nn=False)
    same("utils/knn_idx", kn.idx), same("utils/gathered_values", masked_gather(G(u["values"][0], dev)[None], kn.idx))


def test_grid_reuse_between_calls(dev, oracle):
    """`set_grid_cache(True)`: a second query of the same, unmodified target tensor reuses the grid a previous call
    left in its workspace (pointops_knn_points_idx_reuse levels 1 / 2) -- results bit-identical to the stateless call
    for new queries, repeated queries, ragged lengths (padded rows are rewritten into the new outputs), a self-query,
    clouds without a usable grid and the refined-cell path; an IN-PLACE write to the target, a different K or a new
    target tensor invalidate the cached grid."""
    import pytorch3d_pointops_amd as po
    from pytorch3d_pointops_amd import _C

    a = cases.cloud(3501, (4, 9000, 3))
    b = cases.cloud(3502, (4, 12000, 3))
    b[1] = (b[1] ** np.float32(4.0)).astype(np.float32)  # density gradient: refined cells, box search
    b[3, :, :] = b[3, :1, :]                             # one point 12 000 times: no usable grid for this cloud
    a2 = cases.cloud(3503, (4, 9000, 3))
    l1, l2 = np.array([9000, 8000, 100, 9000]), np.array([12000, 12000, 5000, 12000])
    ta, ta2, tb, tl1, tl2 = G(a, dev), G(a2, dev), G(b, dev), G(l1, dev), G(l2, dev)

    def fresh(p1, p2, la, lb, K):
        po.set_grid_cache(False)
        try:
            return _C.knn_points_idx(p1, p2, la, lb, 2, K, 3)
        finally:
            po.set_grid_cache(True)

    try:
        po.set_grid_cache(True)
        s0 = dict(_C.grid_cache_stats)
        i0, d0 = _C.knn_points_idx(ta, tb, tl1, tl2, 2, 16, 3)       # miss: builds
        i1, d1 = _C.knn_points_idx(ta, tb, tl1, tl2, 2, 16, 3)       # both sides cached: no build pass
        i2, d2 = _C.knn_points_idx(ta2, tb, tl1, tl2, 2, 16, 3)      # new queries: only their sort
        s1 = dict(_C.grid_cache_stats)
        assert (s1["miss"] - s0["miss"], s1["both"] - s0["both"], s1["points"] - s0["points"]) == (1, 1, 1)
        oi, od = oracle.knn_points_idx(a, b, l1, l2, 2, 16)
        for i, d in ((i0, d0), (i1, d1)):
            assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(bits(d.cpu().numpy()), bits(od))
        f2 = fresh(ta2, tb, tl1, tl2, 16)
        assert torch.equal(i2, f2[0]) and torch.equal(d2, f2[1])
        # another K is another grid
        i3, d3 = _C.knn_points_idx(ta, tb, tl1, tl2, 2, 4, 3)
        assert _C.grid_cache_stats["miss"] == s1["miss"] + 1
        f3 = fresh(ta, tb, tl1, tl2, 4)
        assert torch.equal(i3, f3[0]) and torch.equal(d3, f3[1])
        # an in-place write to the target moves its version counter: the cached grid is not used
        _C.knn_points_idx(ta, tb, tl1, tl2, 2, 16, 3)
        tb.mul_(0.5)
        before = dict(_C.grid_cache_stats)
        i4, d4 = _C.knn_points_idx(ta, tb, tl1, tl2, 2, 16, 3)
        assert _C.grid_cache_stats["miss"] == before["miss"] + 1
        f4 = fresh(ta, tb, tl1, tl2, 16)
        assert torch.equal(i4, f4[0]) and torch.equal(d4, f4[1]) and not torch.equal(d4, d0)
        # new lengths tensor for the target (same values elsewhere): a new grid as well
        tl2b = G(np.array([12000, 6000, 5000, 12000]), dev)
        i5, d5 = _C.knn_points_idx(ta, tb, tl1, tl2b, 2, 16, 3)
        f5 = fresh(ta, tb, tl1, tl2b, 16)
        assert torch.equal(i5, f5[0]) and torch.equal(d5, f5[1])
        # self-query, then other queries against the same cloud, then the self-query again
        s_i, s_d = _C.knn_points_idx(tb, tb, tl2, tl2, 2, 8, 3)
        q_i, q_d = _C.knn_points_idx(ta, tb, tl1, tl2, 2, 8, 3)
        r_i, r_d = _C.knn_points_idx(tb, tb, tl2, tl2, 2, 8, 3)
        fs, fq = fresh(tb, tb, tl2, tl2, 8), fresh(ta, tb, tl1, tl2, 8)
        assert torch.equal(s_i, fs[0]) and torch.equal(s_d, fs[1]) and torch.equal(r_i, fs[0]) and torch.equal(r_d, fs[1])
        assert torch.equal(q_i, fq[0]) and torch.equal(q_d, fq[1])
        # long lists (the wave-sort search) reuse the grid the same way
        w0 = _C.knn_points_idx(ta, tb, tl1, tl2, 2, 80, 3)
        w1 = _C.knn_points_idx(ta, tb, tl1, tl2, 2, 80, 3)
        w2 = _C.knn_points_idx(ta2, tb, tl1, tl2, 2, 80, 3)
        fw, fw2 = fresh(ta, tb, tl1, tl2, 80), fresh(ta2, tb, tl1, tl2, 80)
        assert torch.equal(w0[0], fw[0]) and torch.equal(w1[0], fw[0]) and torch.equal(w1[1], fw[1])
        assert torch.equal(w2[0], fw2[0]) and torch.equal(w2[1], fw2[1])
        # the functional API goes through the same cache (chamfer against a fixed target: the target-side grid is kept)
        from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

        x = G(cases.cloud(3504, (2, 30000, 3)), dev).requires_grad_(True)
        y = G(cases.cloud(3505, (2, 30000, 3)), dev)
        before = dict(_C.grid_cache_stats)
        l_a, _ = chamfer_distance(x, y)
        l_b, _ = chamfer_distance(x, y)
        assert torch.equal(l_a, l_b) and _C.grid_cache_stats["both"] > before["both"]
        with torch.no_grad():
            x.add_(0.01)  # an optimiser step: x's grids are rebuilt, y's is reused for the x -> y direction
        mid = dict(_C.grid_cache_stats)
        l_c, _ = chamfer_distance(x, y)
        assert _C.grid_cache_stats["points"] > mid["points"]
        po.set_grid_cache(False)
        l_d, _ = chamfer_distance(x, y)
        assert torch.equal(l_c, l_d)
    finally:
        po.set_grid_cache(False)


def _bruteforce_rows(q, pts, K):
    """(idx, dists) of the K nearest of `pts` for every row of `q` by a dense torch distance table with the kernels'
    expression ((dx*dx + dy*dy) + dz*dz) and lexicographic (dist, idx) order (stable sort)."""
    d = q[:, None, :] - pts[None, :, :]
    d = d * d
    full = (d[..., 0] + d[..., 1]) + d[..., 2]
    val, idx = torch.sort(full, dim=1, stable=True)
    return idx[:, :K], val[:, :K]


@pytest.mark.parametrize("P,K", [(2_000_000, 16), (5_000_000, 8)])
def test_knn_single_huge_cloud(dev, P, K):
    """One cloud of 2 M points (the 21-bit run words) and one of 5 M (the 24-bit run words of knn_grid_d3w.hip: round 2
    refused clouds over 2^20 points and fell back to the all-pairs scan, ~1 s for 2 M x 2 M): self-excluded p1 != p2
    query through the grid, sampled rows against a brute-force torch distance table, and sortedness / range
    properties of every row."""
    from pytorch3d_pointops_amd import _C, synth

    p1 = G(synth.uniform_f32(3601, (1, P, 3)), dev)
    p2 = G(synth.uniform_f32(3602, (1, P, 3)), dev)
    L = torch.full((1,), P, dtype=torch.int64, device=dev)
    assert _C._lib.pointops_knn_uses_grid(1, P, P, 3, K, -1) == 1
    idx, d = _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        idx, d = _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
    torch.cuda.synchronize()
    print(f"knn {P} x {P} K={K}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per call")
    rows = torch.arange(17, P, P // 61, device=dev)
    bi, bd = _bruteforce_rows(p1[0, rows], p2[0], K)
    assert torch.equal(idx[0, rows], bi) and torch.equal(d[0, rows], bd)
    assert bool((d[0, :, 1:] >= d[0, :, :-1]).all()) and bool((idx >= 0).all()) and bool((idx < P).all())
    # every row's distances are the distances to the points it names
    sel = torch.arange(0, P, 1009, device=dev)
    g = p2[0][idx[0, sel]] - p1[0, sel][:, None, :]
    g = g * g
    assert torch.equal((g[..., 0] + g[..., 1]) + g[..., 2], d[0, sel])


def test_knn_grid_wide_run_words_and_batch_slices(dev, oracle, monkeypatch):
    """(a) The 24-bit / 8-bit run words (clouds over 2^21 points) forced on small adversarial clouds -- runs longer than
    255 records must reach the box search -- against the oracle; (b) more clouds than one launch takes (> 32768: the
    batch is searched in slices through one workspace) against the brute-force family."""
    from pytorch3d_pointops_amd import _C

    a = cases.cloud(3701, (3, 6000, 3))
    b = cases.cloud(3702, (3, 20000, 3))
    b[1, :8000] = b[1, :8000] * np.float32(2e-3) + np.float32(0.3)  # a cluster: runs of thousands of records
    b[2] = (b[2] ** np.float32(3.0)).astype(np.float32)
    l1, l2 = np.array([6000, 6000, 777]), np.array([20000, 20000, 15000])
    monkeypatch.setenv("POINTOPS_DEBUG", "grid_big=1")
    for K, norm in ((16, 2), (3, 1), (40, 2)):
        idx, d, st = _C.knn_grid_stats(G(a, dev), G(b, dev), G(l1, dev), G(l2, dev), norm, K)
        oi, od = oracle.knn_points_idx(a, b, l1, l2, norm, K)
        assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(bits(d.cpu().numpy()), bits(od)), (K, norm)
        assert int(st.cpu().numpy()[1, 8]) > 0  # the cluster's queries went to the box search
    monkeypatch.delenv("POINTOPS_DEBUG")
    N = 33000
    p = G(cases.cloud(3703, (N, 48, 3)), dev)
    q = G(cases.cloud(3704, (N, 20, 3)), dev)
    lp = torch.full((N,), 48, dtype=torch.int64, device=dev)
    lq = torch.full((N,), 20, dtype=torch.int64, device=dev)
    lp[::7] = 5
    i3, d3 = _C.knn_points_idx(q, p, lq, lp, 2, 4, 3)
    i2, d2 = _C.knn_points_idx(q, p, lq, lp, 2, 4, 2)
    assert torch.equal(i3, i2) and torch.equal(d3, d2)


@pytest.mark.parametrize("K,norm,D", [(65, 2, 3), (100, 2, 3), (128, 2, 3), (100, 1, 3), (96, 2, 2), (80, 2, 1)])
def test_knn_grid_wave_sort_long_lists(dev, oracle, K, norm, D):
    """64 < K <= 128 through the grid family (knn_grid_wsort.hip: a wave per query, 2048-key sorts of its cube's
    candidates): ragged clouds, a cluster whose cubes hold more records than the kernel streams (those queries take the
    all-pairs list; smaller overflows are sorted chunk by chunk), a cloud shorter than K, duplicated points (ties
    resolved by index), against the oracle and the brute-force family."""
    from pytorch3d_pointops_amd import _C

    p1 = cases.cloud(3801, (4, 1500, D))
    p2 = cases.cloud(3802, (4, 30000, D))
    p2[1, :20000] = p2[1, :20000] * np.float32(3e-3) + np.float32(0.4)  # cluster: cubes with > 16384 records
    p2[2, 1::2] = p2[2, ::2]  # every point twice: ties in every list
    l1 = np.array([1500, 1500, 1500, 40])
    l2 = np.array([30000, 30000, 21000, K - 7])
    args = (G(p1, dev), G(p2, dev), G(l1, dev), G(l2, dev), norm, K)
    idx, d, counts = _C.knn_grid_fallback_counts(*args)
    oi, od = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(d.cpu().numpy()), bits(od))
    counts = counts.cpu().numpy()
    assert 0 < counts[1, 1], counts  # the cluster's neighbourhood exceeds the stream: all-pairs list
    if norm == 2 and D == 3:  # (cells are sized for 3-D Euclidean balls: 0.4 K points)
        assert counts[1, 0] == 0, counts  # every query of the uniform cloud is certified by its radius-1 or -2 cube
    i0, d0 = _C.knn_points_idx(*args, 0)
    assert torch.equal(i0, idx) and torch.equal(d0, d)


def test_knn_long_list_big_cloud(dev):
    """One 300 000-point cloud, K = 100 (round 2: the all-pairs scan, 9e10 pairs): sampled rows against a brute-force
    torch distance table, sortedness of every row."""
    from pytorch3d_pointops_amd import _C, synth

    P, K = 300_000, 100
    p1 = G(synth.uniform_f32(3811, (1, P, 3)), dev)
    p2 = G(synth.uniform_f32(3812, (1, P, 3)), dev)
    L = torch.full((1,), P, dtype=torch.int64, device=dev)
    idx, d = _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        idx, d = _C.knn_points_idx(p1, p2, L, L, 2, K, -1)
    torch.cuda.synchronize()
    print(f"knn {P} x {P} K={K}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per call")
    rows = torch.arange(5, P, P // 97, device=dev)
    bi, bd = _bruteforce_rows(p1[0, rows], p2[0], K)
    assert torch.equal(idx[0, rows], bi) and torch.equal(d[0, rows], bd)
    assert bool((d[0, :, 1:] >= d[0, :, :-1]).all()) and bool((idx >= 0).all()) and bool((idx < P).all())
