#!/usr/bin/env python3
"""Generate tests/golden/*.npz FROM THE REFERENCE ITSELF (build container only).

What runs here: the reference's Python wrappers imported from /root/reference
(`pytorch3d_pointops.functions.*`, `pytorch3d_pointops.structures.Pointclouds`)
on top of the reference's own CPU kernels compiled by oracle/build_ref.py into
oracle/_ref/_C*.so -- nothing from this repository's product or oracle code is
involved in producing the expected outputs.  Inputs come from tests/cases.py
(seeded, regenerable anywhere), so the fixtures hold only outputs (+ digests).

    python tests/golden/make_golden.py            # small fixtures (seconds)
    python tests/golden/make_golden.py --big      # + cfg2-size single-cloud digests (~1 min)

The reference cannot travel to the GPU box; these fixtures are what pins parity there.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.build_ref import build as build_ref, OUT_DIR as REF_DIR  # noqa: E402

build_ref(verbose=True)
sys.path.insert(0, "/root/reference")
import pytorch3d_pointops  # noqa: E402  (reference package; __init__ only sets __version__)

pytorch3d_pointops.__path__.append(REF_DIR)  # lets `from pytorch3d_pointops import _C` find oracle/_ref/_C*.so
from pytorch3d_pointops import _C as ref_C  # noqa: E402
from pytorch3d_pointops.functions import (  # noqa: E402
    ball_query, knn_gather, knn_points, packed_to_padded, padded_to_packed, sample_farthest_points, masked_gather)
from pytorch3d_pointops.functions.chamfer import chamfer_distance  # noqa: E402
from pytorch3d_pointops.functions.sample_farthest_points import sample_farthest_points_naive  # noqa: E402
from pytorch3d_pointops.structures import Pointclouds  # noqa: E402

import cases  # noqa: E402
from pytorch3d_pointops_amd import synth  # noqa: E402  (input generator only)


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if dtype is None else t.to(dtype)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path)} B, {len(arrays)} arrays)")


def gen_knn():
    out = {}
    for name, c in cases.knn_cases().items():
        r = knn_points(T(c["p1"]), T(c["p2"]), T(c["l1"], torch.int64), T(c["l2"], torch.int64),
                       norm=c["norm"], K=c["K"], return_nn=True)
        out[name + "/dists"] = r.dists.numpy()
        out[name + "/idx"] = r.idx.numpy().astype(np.int32)
        out[name + "/knn"] = r.knn.numpy()
        # the raw operator (no python-side sort) must agree with the wrapper on CPU
        i2, d2 = ref_C.knn_points_idx(T(c["p1"]), T(c["p2"]), T(c["l1"], torch.int64), T(c["l2"], torch.int64),
                                      c["norm"], c["K"], -1)
        assert torch.equal(i2, r.idx) and torch.equal(d2, r.dists), name
    save("knn", **out)


def gen_knn_backward():
    out = {}
    for name, c in cases.knn_backward_cases().items():
        p1 = T(c["p1"]).requires_grad_(True)
        p2 = T(c["p2"]).requires_grad_(True)
        r = knn_points(p1, p2, T(c["l1"], torch.int64), T(c["l2"], torch.int64), norm=c["norm"], K=c["K"])
        g = T(cases.grad_for(name, tuple(r.dists.shape)))
        r.dists.backward(g)
        out[name + "/grad_p1"] = p1.grad.numpy()
        out[name + "/grad_p2"] = p2.grad.numpy()
    save("knn_backward", **out)


def gen_ball_query():
    out = {}
    for name, c in cases.ball_query_cases().items():
        p1 = T(c["p1"]).requires_grad_(True)
        p2 = T(c["p2"]).requires_grad_(True)
        r = ball_query(p1, p2, T(c["l1"], torch.int64), T(c["l2"], torch.int64), K=c["K"], radius=c["radius"],
                       return_nn=True)
        out[name + "/dists"] = r.dists.detach().numpy()
        out[name + "/idx"] = r.idx.numpy().astype(np.int32)
        out[name + "/knn"] = r.knn.detach().numpy()
        g = T(cases.grad_for("bq" + name, tuple(r.dists.shape)))
        (r.dists * g).sum().backward()
        out[name + "/grad_p1"] = p1.grad.numpy()
        out[name + "/grad_p2"] = p2.grad.numpy()
    save("ball_query", **out)


def gen_fps():
    out = {}
    for name, c in cases.fps_cases().items():
        pts = T(c["points"])
        idx = ref_C.sample_farthest_points(pts, T(c["lengths"], torch.int64), T(c["K"], torch.int64),
                                           T(c["start"], torch.int64))
        out[name + "/idx"] = idx.numpy().astype(np.int32)
        out[name + "/points"] = masked_gather(pts, idx).numpy()
        if (c["start"] == 0).all():
            # wrapper path (start 0) and the reference's pure-torch naive variant must agree
            sp, si = sample_farthest_points(pts, T(c["lengths"], torch.int64), T(c["K"], torch.int64))
            assert torch.equal(si, idx), name
            if name != "all_equal":
                _, ni = sample_farthest_points_naive(pts, T(c["lengths"], torch.int64), T(c["K"], torch.int64))
                assert torch.equal(ni, idx), name
    save("fps", **out)


def gen_packed():
    out = {}
    for name, c in cases.packed_cases().items():
        x, first, F = cases.packed_inputs(c)
        xt = T(x).requires_grad_(True)
        arg = xt[:, 0] if c["D"] == 1 else xt
        padded = packed_to_padded(arg, T(first), int(c["max_size"]))
        out[name + "/padded"] = padded.detach().numpy()
        g = T(cases.grad_for("pp" + name, tuple(padded.shape)))
        (padded * g).sum().backward()
        out[name + "/grad_packed"] = xt.grad.numpy()
        back = padded_to_packed(padded.detach(), T(first), F)
        out[name + "/roundtrip"] = back.numpy()
        # padded -> packed of an arbitrary padded tensor (padding NOT zero) + its gradient
        pt = T(cases.grad_for("pq" + name, tuple(padded.shape))).requires_grad_(True)
        pk = padded_to_packed(pt, T(first), F)
        out[name + "/packed_of_g"] = pk.detach().numpy()
        pk.sum().backward()
        out[name + "/grad_padded_ones"] = pt.grad.numpy()
    # 3-D trailing dims through the wrapper
    x = cases.cloud(410, (15, 2, 3))
    first = np.array([0, 5, 5, 12], np.int64)
    p3 = packed_to_padded(T(x), T(first), 7)
    out["trailing/padded"] = p3.numpy()
    out["trailing/roundtrip"] = padded_to_packed(p3, T(first), 15).numpy()
    save("packed_padded", **out)


def gen_gather():
    out = {}
    x = T(cases.cloud(601, (2, 50, 5)))
    idx = T(synth.randint(602, 0, 49, (2, 30, 4)))
    lengths = T(np.array([50, 2]), torch.int64)
    xr = x.clone().requires_grad_(True)
    o = knn_gather(xr, idx, lengths)
    out["knn_gather/out"] = o.detach().numpy()
    g = T(cases.grad_for("kg", tuple(o.shape)))
    (o * g).sum().backward()
    out["knn_gather/grad_x"] = xr.grad.numpy()
    midx = idx.clone()
    midx[0, ::3, 1] = -1
    midx[1, :, 3] = -1
    xr2 = x.clone().requires_grad_(True)
    o2 = masked_gather(xr2, midx)
    out["masked_gather3/out"] = o2.detach().numpy()
    (o2 * g).sum().backward()
    out["masked_gather3/grad_x"] = xr2.grad.numpy()
    m2 = midx[:, :, 1].contiguous()
    out["masked_gather2/out"] = masked_gather(x, m2).numpy()
    save("gather", **out)


def gen_chamfer():
    out = {}
    inp = cases.chamfer_inputs()
    for v in cases.chamfer_variants():
        key = cases.variant_key(v)
        x = T(inp["x"]).requires_grad_(True)
        y = T(inp["y"]).requires_grad_(True)
        xn = T(inp["xn"]).requires_grad_(True)
        yn = T(inp["yn"]).requires_grad_(True)
        kw = dict(x_lengths=T(inp["xl"], torch.int64), y_lengths=T(inp["yl"], torch.int64),
                  batch_reduction=v["batch_reduction"], point_reduction=v["point_reduction"], norm=v["norm"],
                  single_directional=v["single_directional"], abs_cosine=v["abs_cosine"])
        if v["use_weights"]:
            kw["weights"] = T(inp["w"])
        if v["features"]:
            kw.update(x_features={"normals": xn}, y_features={"normals": yn}, feature_names=["normals"])
        loss, lf = chamfer_distance(x, y, **kw)
        flat = []

        def put(tag, t):
            if isinstance(t, tuple):
                for i, tt in enumerate(t):
                    put(f"{tag}{i}", tt)
            elif t is not None:
                out[f"{key}/{tag}"] = t.detach().numpy()
                flat.append(t)

        put("loss", loss)
        if lf is not None:
            put("lossf", lf["normals"])
        total = sum(t.sum() for t in flat)
        total.backward()
        out[f"{key}/grad_x"] = x.grad.numpy()
        out[f"{key}/grad_y"] = y.grad.numpy()
        if v["features"]:
            out[f"{key}/grad_xn"] = xn.grad.numpy()
            out[f"{key}/grad_yn"] = yn.grad.numpy()
    # Pointclouds inputs (ragged lists + "normals" feature) must equal the tensor path
    xl, yl = inp["xl"], inp["yl"]
    pc_x = Pointclouds([T(inp["x"][n, : xl[n]]) for n in range(3)],
                       features={"normals": [T(inp["xn"][n, : xl[n]]) for n in range(3)]})
    pc_y = Pointclouds([T(inp["y"][n, : yl[n]]) for n in range(3)],
                       features={"normals": [T(inp["yn"][n, : yl[n]]) for n in range(3)]})
    loss, lf = chamfer_distance(pc_x, pc_y, feature_names=["normals"])
    out["pointclouds/loss"] = loss.numpy()
    out["pointclouds/lossf"] = lf["normals"].numpy()
    save("chamfer", **out)


def gen_covariances():
    """get_point_covariances / wmean (functions/utils.py:68-153): forward values and the gradient w.r.t. the
    points through knn_gather and the covariance arithmetic."""
    from pytorch3d_pointops.functions.utils import get_point_covariances, wmean

    out = {}
    for name, (N, P, D, K, lens) in {"d3_k8": (2, 120, 3, 8, [120, 57]), "d2_k5": (1, 90, 2, 5, [90]),
                                     "d3_k16_ties": (1, 150, 3, 16, [150])}.items():
        pts = cases.lattice(811, N, P, D, levels=6) if "ties" in name else cases.cloud(810 + D + K, (N, P, D))
        x = T(pts).requires_grad_(True)
        cov, knn = get_point_covariances(x, T(np.array(lens), torch.int64), K)
        out[name + "/cov"] = cov.detach().numpy()
        out[name + "/knn"] = knn.detach().numpy()
        gc = T(cases.grad_for("cov" + name, tuple(cov.shape)))
        gk = T(cases.grad_for("knn" + name, tuple(knn.shape)))
        ((cov * gc).sum() + (knn * gk).sum()).backward()
        out[name + "/grad_points"] = x.grad.numpy()
    xw = T(cases.cloud(820, (2, 40, 3)))
    w = T(synth.uniform_f32(821, (2, 40)))
    out["wmean/weighted"] = wmean(xw, w).numpy()
    out["wmean/plain"] = wmean(xw).numpy()
    save("covariances", **out)


def gen_sample_pdf():
    from pytorch3d_pointops.functions.sample_pdf import sample_pdf as ref_sample_pdf

    out = {}
    for name, c in cases.sample_pdf_cases().items():
        o = T(c["u"].copy())
        ref_C.sample_pdf(T(c["bins"]), T(c["weights"]), o, c["eps"])
        out[name + "/samples"] = o.numpy()
    # wrapper, deterministic quantiles, batch shape (2, 4)
    c = cases.sample_pdf_cases()["b8_64x128"]
    det = ref_sample_pdf(T(c["bins"]).reshape(2, 4, -1), T(c["weights"]).reshape(2, 4, -1), 40, det=True)
    out["wrapper_det/samples"] = det.numpy()
    save("sample_pdf", **out)


def gen_examples():
    """The call patterns of the reference's own examples/knn_on_pointclouds.py (:24,35,78-88,130-190) at its
    sizes: a Pointclouds batch of a 1500-point sphere shell and an 800-point ellipsoid shell with "normals"
    and "colors" features; self-KNN K=10 on the padded batch with per-cloud lengths, a 200 x 800 cross query
    K=5, knn_gather of normals / colors through the neighbour table, inverse-distance interpolation.  The
    script draws its clouds from torch.rand; the fixture stores the (synth-generated) inputs next to the
    reference's outputs so that nothing depends on an RNG or a libm."""
    from pytorch3d_pointops.structures import Pointclouds as RefPointclouds

    inp = cases.example_clouds()
    pc = RefPointclouds(points=[T(a) for a in inp["points"]],
                        features={"normals": [T(a) for a in inp["normals"]], "colors": [T(a) for a in inp["colors"]]})
    out = {f"in/points{i}": a for i, a in enumerate(inp["points"])}
    out.update({f"in/normals{i}": a for i, a in enumerate(inp["normals"])})
    out.update({f"in/colors{i}": a for i, a in enumerate(inp["colors"])})
    padded, lens = pc.points_padded(), pc.num_points_per_cloud()
    out["self/padded"] = padded.numpy()
    r = knn_points(padded, padded, lengths1=lens, lengths2=lens, K=10, return_nn=True)  # :81-88
    out["self/dists"], out["self/idx"], out["self/knn"] = r.dists.numpy(), r.idx.numpy().astype(np.int32), r.knn.numpy()
    q = T(inp["points"][0][:200])[None]  # :130-149
    t = T(inp["points"][1])[None]
    c = knn_points(p1=q, p2=t, K=5, return_nn=False)
    out["cross/dists"], out["cross/idx"] = c.dists.numpy(), c.idx.numpy().astype(np.int32)
    gn = knn_gather(T(inp["normals"][1])[None], c.idx)[0]  # :155-156
    gc = knn_gather(T(inp["colors"][1])[None], c.idx)[0]
    out["cross/gathered_normals"], out["cross/gathered_colors"] = gn.numpy(), gc.numpy()
    w = 1.0 / (torch.sqrt(c.dists[0]) + 1e-8)  # :165-178
    w = w / w.sum(dim=1, keepdim=True)
    out["cross/interp_normals"] = torch.nn.functional.normalize((gn * w.unsqueeze(-1)).sum(dim=1), p=2, dim=1).numpy()
    out["cross/interp_colors"] = (gc * w.unsqueeze(-1)).sum(dim=1).numpy()
    save("examples", **out)


def gen_examples2():
    """The call patterns of the reference's other self-checking example scripts at their own sizes (inputs:
    cases.example2_inputs(), regenerated by the tests):
      examples/ball_query_on_pointclouds.py:50-125   self ball query of a Pointclouds batch (K=50, r=0.3, return_nn)
                                                     and a 10x10x10 lattice (K=30, r=0.25) next to knn_points K=10
      examples/fps_on_pointclouds.py:66-215          single cloud / ragged batch with per-cloud K and random starts
                                                     (torch.manual_seed before each call), optimized vs naive, 2-D
      examples/chamfer_loss.py:13-89                 tensors with two feature sets, Pointclouds inputs, single direction
      examples/packed_to_padded_on_pointclouds.py:67-124   packed <-> padded round trips of points and features
      examples/utils_on_pointclouds.py:69-237        get_point_covariances K=16, wmean, knn_points K=8 + masked_gather
    Bit-exact outputs larger than a few KB are stored as sha256 digests plus every 37th row."""
    from pytorch3d_pointops.functions.utils import get_point_covariances, wmean
    from pytorch3d_pointops.structures import Pointclouds as RefPointclouds

    inp = cases.example2_inputs()
    out, meta = {}, {}

    def put(tag, t, exact=True):
        a = t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        if a.dtype == np.int64:
            a = a.astype(np.int32)
        if exact and a.nbytes > 8192:
            meta[tag] = dict(sha256=hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest(),
                             shape=list(a.shape), dtype=str(a.dtype))
            flat = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a
            out[tag + "/rows"] = flat[::37]
        else:
            out[tag] = a

    # ---- ball query
    b = inp["ball"]
    pc = RefPointclouds(points=[T(a) for a in b["points"]])
    pad, lens = pc.points_padded(), pc.num_points_per_cloud()
    r = ball_query(p1=pad, p2=pad, lengths1=lens, lengths2=lens, K=50, radius=0.3, return_nn=True)
    put("ball/dists", r.dists), put("ball/idx", r.idx), put("ball/knn", r.knn)
    lat = T(b["lattice"])[None]
    r = ball_query(p1=lat, p2=lat, K=30, radius=0.25, return_nn=False)
    k = knn_points(p1=lat, p2=lat, K=10, return_nn=False)
    put("lattice/ball_dists", r.dists), put("lattice/ball_idx", r.idx)
    put("lattice/knn_dists", k.dists), put("lattice/knn_idx", k.idx)
    # ---- FPS
    f = inp["fps"]
    torch.manual_seed(123)
    sp, si = sample_farthest_points(T(f["single"])[None], K=50, random_start_point=True)
    put("fps/single_idx", si), put("fps/single_points", sp)
    pc = RefPointclouds(points=[T(a) for a in f["points"]])
    torch.manual_seed(42)
    sp, si = sample_farthest_points(pc.points_padded(), lengths=pc.num_points_per_cloud(), K=[100, 80, 150],
                                    random_start_point=True)
    put("fps/batch_idx", si), put("fps/batch_points", sp)
    so, io = sample_farthest_points(T(f["compare"]), K=200, random_start_point=False)
    sn, in_ = sample_farthest_points_naive(T(f["compare"]), K=200, random_start_point=False)
    assert torch.equal(io, in_)
    put("fps/compare_idx", io)
    torch.manual_seed(7)
    sp, si = sample_farthest_points(T(f["points"][0])[None], K=100, random_start_point=True)
    put("fps/colors_idx", si), put("fps/colors", masked_gather(T(f["colors0"])[None], si))
    sn, in_ = sample_farthest_points_naive(T(f["circles"])[None], K=50, random_start_point=False)
    put("fps/circles_idx", in_)
    # ---- chamfer
    c = {k: T(v) for k, v in inp["chamfer"].items()}
    l, lf = chamfer_distance(c["p1"], c["p2"], x_features={"normals": c["n1"], "colors": c["c1"]},
                             y_features={"normals": c["n2"], "colors": c["c2"]}, feature_names=["normals", "colors"])
    out["chamfer/tensor"] = np.array([float(l), float(lf["normals"]), float(lf["colors"])], np.float64)
    pc1 = RefPointclouds(points=list(c["p1"]), features={"normals": list(c["n1"]), "colors": list(c["c1"])})
    pc2 = RefPointclouds(points=list(c["p2"]), features={"normals": list(c["n2"]), "colors": list(c["c2"])})
    l, lf = chamfer_distance(pc1, pc2, feature_names=["normals", "colors"])
    out["chamfer/pointclouds"] = np.array([float(l), float(lf["normals"]), float(lf["colors"])], np.float64)
    l, lf = chamfer_distance(pc1, pc2, feature_names=["normals"], single_directional=True)
    out["chamfer/single"] = np.array([float(l), float(lf["normals"])], np.float64)
    # ---- packed <-> padded
    k = inp["packed"]
    pc = RefPointclouds(points=[T(a) for a in k["points"]], features={"intensities": [T(a) for a in k["intensities"]]})
    lens = pc.num_points_per_cloud()
    first = torch.cat([torch.tensor([0]), lens.cumsum(0)[:-1]])
    pp, ip = pc.points_packed(), pc.get_features_packed("intensities")
    pad, ipad = packed_to_padded(pp, first, int(lens.max())), packed_to_padded(ip, first, int(lens.max()))
    put("packed/points_packed", pp), put("packed/points_padded", pad), put("packed/intensities_padded", ipad)
    put("packed/points_repacked", padded_to_packed(pad, first, int(lens.sum())))
    put("packed/intensities_repacked", padded_to_packed(ipad, first, int(lens.sum())))
    pc = RefPointclouds(points=[T(a) for a in k["var_points"]], features={"features": [T(a) for a in k["var_features"]]})
    lens = pc.num_points_per_cloud()
    first = torch.cat([torch.tensor([0]), lens.cumsum(0)[:-1]])
    fpad = packed_to_padded(pc.get_features_packed("features"), first, int(lens.max()))
    put("packed/var_features_padded", fpad)
    put("packed/var_features_repacked", padded_to_packed(fpad, first, int(lens.sum())))
    # ---- utils
    u = inp["utils"]
    pc = RefPointclouds(points=[T(a) for a in u["points"]])
    cov, nn = get_point_covariances(pc.points_padded(), pc.num_points_per_cloud(), 16)
    out["utils/cov"] = cov.numpy()[:, ::5]
    put("utils/cov_knn", nn)
    out["utils/wmean"] = np.stack([wmean(T(p), T(w).squeeze(), dim=0, keepdim=False).numpy()
                                   for p, w in zip(u["points"], u["weights"])])
    kn = knn_points(T(u["points"][0])[None], T(u["points"][0])[None], K=8, return_nn=False)
    put("utils/knn_idx", kn.idx), put("utils/gathered_values", masked_gather(T(u["values"][0])[None], kn.idx))
    save("examples2", **out)
    with open(os.path.join(HERE, "examples2_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print("wrote examples2_meta.json")


def gen_pointclouds():
    """The container calls of examples/pointclouds.py:12-174 (construction with named features, the three views, one
    cloud, update_padded with and without new features) plus the module-level helpers its callers use (offset, scale,
    bounding boxes, scene join, inside_box, extend, split) through the REFERENCE's Pointclouds: host-side bookkeeping,
    so the fixture is checked on the CPU (tests/test_boundary_cpu.py)."""
    from pytorch3d_pointops.structures import point_structure as ps

    inp = cases.example_clouds()
    pc = ps.Pointclouds(points=[T(a) for a in inp["points"]],
                        features={"normals": [T(a) for a in inp["normals"]], "colors": [T(a) for a in inp["colors"]]})
    out = {"padded": pc.points_padded().numpy(), "packed": pc.points_packed().numpy(),
           "normals_padded": pc.get_features_padded("normals").numpy(),
           "colors_packed": pc.get_features_packed("colors").numpy(),
           "first_idx": pc.cloud_to_packed_first_idx().numpy(), "packed_to_cloud": pc.packed_to_cloud_idx().numpy(),
           "padded_to_packed": pc.padded_to_packed_idx().numpy()}
    pts, feats = pc.get_cloud(1)
    out["cloud1_points"], out["cloud1_colors"] = pts.numpy(), feats["colors"].numpy()
    new_pts = pc.points_padded() * 2.0 + 1.0
    up = pc.update_padded(new_pts)
    out["update_packed"], out["update_normals_packed"] = up.points_packed().numpy(), up.get_features_packed("normals").numpy()
    up2 = pc.update_padded(new_pts, new_features_padded={"normals": pc.get_features_padded("normals") * -1.0})
    out["update2_normals_list1"] = up2.get_features_list("normals")[1].numpy()
    out["update2_names"] = np.array(sorted(up2.features_list().keys()))
    off = ps.offset(pc, torch.tensor([0.5, -1.0, 2.0]))
    out["offset_padded"] = off.points_padded().numpy()
    sc = ps.scale(pc, torch.tensor([2.0, 0.25]))
    out["scale_packed"], out["scale_padded"] = sc.points_packed().numpy(), sc.points_padded().numpy()
    out["bboxes"] = ps.get_bounding_boxes(pc).numpy()
    scene = ps.join_pointclouds_as_scene(pc)
    out["scene_padded"], out["scene_colors"] = scene.points_padded().numpy(), scene.get_features_padded("colors").numpy()
    box = torch.tensor([[[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]], [[0.0, -1.0, -1.0], [2.0, 1.0, 1.0]]])
    out["inside_box"] = pc.inside_box(box).numpy()
    ext = pc.extend(2)
    out["extend_lengths"] = ext.num_points_per_cloud().numpy()
    out["extend_padded"] = ext.points_padded().numpy()
    parts = ext.split([1, 3])
    out["split1_lengths"] = parts[1].num_points_per_cloud().numpy()
    out["split1_normals_packed"] = parts[1].get_features_packed("normals").numpy()
    save("pointclouds_api", **out)


def gen_big():
    """cfg2-size single clouds: digests + sampled rows (the full idx would be 8 MB per cloud)."""
    meta = {}
    arrays = {}
    for tag, seed1, seed2, P, K in (("cfg2_cloud", 7001, 7002, 65536, 16),):
        p1 = cases.cloud(seed1, (1, P, 3))
        p2 = cases.cloud(seed2, (1, P, 3))
        L = torch.tensor([P], dtype=torch.int64)
        idx, d = ref_C.knn_points_idx(T(p1), T(p2), L, L, 2, K, -1)
        idx32 = idx.numpy().astype(np.int32)
        meta[tag] = dict(seed1=seed1, seed2=seed2, P=P, K=K,
                         idx_sha256=hashlib.sha256(idx32.tobytes()).hexdigest(),
                         dists_sha256=hashlib.sha256(d.numpy().tobytes()).hexdigest())
        rows = np.arange(0, P, 257)
        arrays[tag + "/rows"] = rows
        arrays[tag + "/idx_rows"] = idx32[0, rows]
        arrays[tag + "/dists_rows"] = d.numpy()[0, rows]
    # FPS at cfg3's per-cloud size
    P, K = 131072, 1024
    pts = cases.cloud(7003, (1, P, 3))
    fi = ref_C.sample_farthest_points(T(pts), torch.tensor([P]), torch.tensor([K]), torch.tensor([0]))
    arrays["cfg3_fps/idx"] = fi.numpy().astype(np.int32)
    meta["cfg3_fps"] = dict(seed=7003, P=P, K=K)
    # ball query at cfg3's per-cloud size, first 2048 queries
    Q = 2048
    bi, bd = ref_C.ball_query(T(pts[:, :Q]), T(pts), torch.tensor([Q]), torch.tensor([P]), 32, 0.2)
    meta["cfg3_bq"] = dict(seed=7003, P=P, Q=Q, K=32, radius=0.2,
                           idx_sha256=hashlib.sha256(bi.numpy().astype(np.int32).tobytes()).hexdigest(),
                           dists_sha256=hashlib.sha256(bd.numpy().tobytes()).hexdigest())
    arrays["cfg3_bq/idx_rows"] = bi.numpy().astype(np.int32)[0, ::64]
    arrays["cfg3_bq/dists_rows"] = bd.numpy()[0, ::64]
    save("big", **arrays)
    with open(os.path.join(HERE, "big_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote big_meta.json")


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--only" in sys.argv:  # regenerate one fixture: --only covariances
        globals()["gen_" + sys.argv[sys.argv.index("--only") + 1]]()
        sys.exit(0)
    gen_knn()
    gen_knn_backward()
    gen_ball_query()
    gen_fps()
    gen_packed()
    gen_gather()
    gen_chamfer()
    gen_sample_pdf()
    gen_covariances()
    gen_examples()
    gen_examples2()
    gen_pointclouds()
    if "--big" in sys.argv:
        gen_big()
