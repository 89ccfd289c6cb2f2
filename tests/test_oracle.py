"""CPU suite: the oracle (plain-C restatement) against the golden vectors produced by
the reference itself (tests/golden/make_golden.py), bit for bit.  Also cross-checks
oracle vs oracle/_ref (the compiled reference kernels) when that file is present."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases
from conftest import GOLDEN, bits, load_golden


@pytest.mark.parametrize("name", sorted(cases.knn_cases()))
def test_knn_oracle_matches_reference_golden(oracle, name):
    g = load_golden("knn")
    c = cases.knn_cases()[name]
    idx, d = oracle.knn_points_idx(c["p1"], c["p2"], c["l1"], c["l2"], c["norm"], c["K"])
    assert np.array_equal(idx, g[name + "/idx"].astype(np.int64))
    assert np.array_equal(bits(d), bits(g[name + "/dists"]))
    from oracle.oracle import knn_gather

    assert np.array_equal(knn_gather(c["p2"], idx, c["l2"]), g[name + "/knn"])


@pytest.mark.parametrize("name", sorted(cases.knn_backward_cases()))
def test_knn_backward_oracle(oracle, name):
    g = load_golden("knn_backward")
    gk = load_golden("knn")
    c = cases.knn_backward_cases()[name]
    idx = gk[name + "/idx"].astype(np.int64)
    grad = cases.grad_for(name, idx.shape)
    g1, g2 = oracle.knn_points_backward(c["p1"], c["p2"], c["l1"], c["l2"], idx, c["norm"], grad)
    assert np.array_equal(bits(g1), bits(g[name + "/grad_p1"]))
    assert np.array_equal(bits(g2), bits(g[name + "/grad_p2"]))


@pytest.mark.parametrize("name", sorted(cases.ball_query_cases()))
def test_ball_query_oracle(oracle, name):
    g = load_golden("ball_query")
    c = cases.ball_query_cases()[name]
    idx, d = oracle.ball_query(c["p1"], c["p2"], c["l1"], c["l2"], c["K"], c["radius"])
    assert np.array_equal(idx, g[name + "/idx"].astype(np.int64))
    assert np.array_equal(bits(d), bits(g[name + "/dists"]))
    from oracle.oracle import masked_gather

    assert np.array_equal(masked_gather(c["p2"], idx), g[name + "/knn"])
    grad = cases.grad_for("bq" + name, idx.shape)
    g1, g2 = oracle.knn_points_backward(c["p1"], c["p2"], c["l1"], c["l2"], idx, 2, grad)
    assert np.array_equal(bits(g1), bits(g[name + "/grad_p1"]))
    assert np.array_equal(bits(g2), bits(g[name + "/grad_p2"]))


@pytest.mark.parametrize("name", sorted(cases.fps_cases()))
def test_fps_oracle(oracle, name):
    g = load_golden("fps")
    c = cases.fps_cases()[name]
    idx = oracle.sample_farthest_points(c["points"], c["lengths"], c["K"], c["start"])
    assert np.array_equal(idx, g[name + "/idx"].astype(np.int64))
    from oracle.oracle import masked_gather

    assert np.array_equal(masked_gather(c["points"], idx), g[name + "/points"])


@pytest.mark.parametrize("name", sorted(cases.packed_cases()))
def test_packed_padded_oracle(oracle, name):
    g = load_golden("packed_padded")
    c = cases.packed_cases()[name]
    x, first, F = cases.packed_inputs(c)
    padded = oracle.packed_to_padded(x, first, int(c["max_size"]))
    want = g[name + "/padded"]
    assert np.array_equal(padded.reshape(want.shape), want)
    assert np.array_equal(oracle.padded_to_packed(padded, first, F), g[name + "/roundtrip"].reshape(F, -1))
    # backward of packed_to_padded == padded_to_packed of the upstream gradient
    gp = cases.grad_for("pp" + name, want.shape).reshape(padded.shape)
    assert np.array_equal(oracle.padded_to_packed(gp, first, F), g[name + "/grad_packed"])
    gq = cases.grad_for("pq" + name, want.shape).reshape(padded.shape)
    assert np.array_equal(oracle.padded_to_packed(gq, first, F), g[name + "/packed_of_g"].reshape(F, -1))
    ones = oracle.packed_to_padded(np.ones((F, c["D"]), np.float32), first, int(c["max_size"]))
    assert np.array_equal(ones.reshape(want.shape), g[name + "/grad_padded_ones"])


def test_oracle_vs_compiled_reference_random(oracle, ref_oracle):
    """Extra pin: oracle == oracle/_ref on fresh seeded inputs (sizes beyond the fixtures)."""
    p1 = cases.cloud(801, (2, 700, 3))
    p2 = cases.cloud(802, (2, 2000, 3))
    l1, l2 = np.array([700, 512]), np.array([2000, 1777])
    for norm in (1, 2):
        for K in (1, 16):
            a = oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
            b = ref_oracle.knn_points_idx(p1, p2, l1, l2, norm, K)
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    a = oracle.ball_query(p1, p2, l1, l2, 32, 0.2)
    b = ref_oracle.ball_query(p1, p2, l1, l2, 32, 0.2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    K = np.array([300, 100])
    s = np.array([5, 0])
    assert np.array_equal(oracle.sample_farthest_points(p2, l2, K, s), ref_oracle.sample_farthest_points(p2, l2, K, s))


def test_big_golden_rows_oracle(oracle):
    """cfg2-size cloud: the oracle reproduces sampled rows of the reference run (full digest
    is checked on the GPU path; a full oracle run takes ~30 s and is left to bench.py)."""
    meta_path = os.path.join(GOLDEN, "big_meta.json")
    if not os.path.exists(meta_path):
        pytest.skip("big fixtures not generated")
    meta = json.load(open(meta_path))["cfg2_cloud"]
    g = load_golden("big")
    rows = g["cfg2_cloud/rows"][:24]
    P, K = meta["P"], meta["K"]
    p1 = cases.cloud(meta["seed1"], (1, P, 3))
    p2 = cases.cloud(meta["seed2"], (1, P, 3))
    idx, d = oracle.knn_points_idx(p1[:, rows], p2, np.array([len(rows)]), np.array([P]), 2, K)
    assert np.array_equal(idx[0], g["cfg2_cloud/idx_rows"][: len(rows)].astype(np.int64))
    assert np.array_equal(bits(d[0]), bits(g["cfg2_cloud/dists_rows"][: len(rows)]))


@pytest.mark.parametrize("name", sorted(cases.sample_pdf_cases()))
def test_sample_pdf_oracle(oracle, name):
    g = load_golden("sample_pdf")
    c = cases.sample_pdf_cases()[name]
    got = oracle.sample_pdf(c["bins"], c["weights"], c["u"], c["eps"])
    assert np.array_equal(bits(got), bits(g[name + "/samples"]))
