"""CPU suite: the C-ABI library loads, exports every symbol include/pointops_amd.h
declares, and the host-side mirror of the reference API validates arguments with
the reference's error types/messages.  No compute calls (there is no GPU here and
the product has no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pointops_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pointops_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from pytorch3d_pointops_amd import _C

    lib = ctypes.CDLL(_C.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 14
    for name in decl:
        assert hasattr(lib, name), f"libpointops_amd.so lacks {name}"
    assert sorted(_C.exported_symbols()) == decl  # the ctypes table binds exactly the header
    assert lib.pointops_abi_version() == 1
    lib.pointops_target_arch.restype = ctypes.c_char_p
    assert lib.pointops_target_arch() == b"gfx950"


def test_code_object_is_gfx950():
    from pytorch3d_pointops_amd import _C

    blob = open(_C.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_no_cpu_fallback():
    from pytorch3d_pointops_amd.functions import knn_points, ball_query, sample_farthest_points, packed_to_padded

    p = torch.rand(1, 8, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        knn_points(p, p, K=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ball_query(p, p, K=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sample_farthest_points(p, K=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        packed_to_padded(torch.rand(8, 3), torch.tensor([0, 4]), 4)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "pytorch3d_pointops_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{fn} mentions the oracle"


def test_knn_validation_errors():
    from pytorch3d_pointops_amd.functions import knn_points, knn_gather

    with pytest.raises(ValueError, match="same batch dimension"):
        knn_points(torch.rand(2, 4, 3), torch.rand(3, 4, 3))
    with pytest.raises(ValueError, match="same point dimension"):
        knn_points(torch.rand(2, 4, 3), torch.rand(2, 4, 2))
    with pytest.raises(ValueError, match="same batch dimension"):
        knn_gather(torch.rand(2, 4, 3), torch.zeros(3, 4, 1, dtype=torch.int64))


def test_ball_query_validation_errors():
    from pytorch3d_pointops_amd.functions import ball_query

    with pytest.raises(ValueError, match="same batch dimension"):
        ball_query(torch.rand(2, 4, 3), torch.rand(3, 4, 3))
    with pytest.raises(ValueError, match="same point dimension"):
        ball_query(torch.rand(2, 4, 3), torch.rand(2, 4, 2))


def test_fps_validation_errors():
    from pytorch3d_pointops_amd.functions import sample_farthest_points

    p = torch.rand(2, 10, 3)
    with pytest.raises(ValueError, match="same batch dimension"):
        sample_farthest_points(p, lengths=torch.tensor([10, 10, 10]))
    with pytest.raises(ValueError, match="too large"):
        sample_farthest_points(p, lengths=torch.tensor([10, 11]))
    with pytest.raises(ValueError, match="K and points"):
        sample_farthest_points(p, K=[1, 2, 3])


def test_packed_validation_errors():
    from pytorch3d_pointops_amd.functions.packed_to_padded import _RaggedCopyFn, packed_to_padded, padded_to_packed

    f = torch.tensor([0, 2])
    # the reference's texts (functions/packed_to_padded.py:38-47, :130-139), raised by the shared autograd node
    with pytest.raises(ValueError, match="input can only be 2-dimensional."):
        _RaggedCopyFn.apply(torch.rand(4), f, 2, True)
    with pytest.raises(ValueError, match="first_idxs can only be 1-dimensional."):
        _RaggedCopyFn.apply(torch.rand(4, 3), f[None], 2, True)
    with pytest.raises(ValueError, match="input has to be of type torch.float32."):
        _RaggedCopyFn.apply(torch.rand(4, 3).double(), f, 2, True)
    with pytest.raises(ValueError, match="first_idxs has to be of type torch.int64."):
        _RaggedCopyFn.apply(torch.rand(4, 3), f.int(), 2, True)
    with pytest.raises(ValueError, match="max_size has to be int."):
        _RaggedCopyFn.apply(torch.rand(4, 3), f, 2.0, True)
    with pytest.raises(ValueError, match="input can only be 3-dimensional."):
        _RaggedCopyFn.apply(torch.rand(4, 3), f, 4, False)
    # and through the public wrappers
    with pytest.raises(ValueError, match="float32"):
        packed_to_padded(torch.rand(4, 3).double(), f, 2)
    with pytest.raises(ValueError, match="int64"):
        padded_to_packed(torch.rand(2, 2, 3), f.int(), 4)


def test_chamfer_validation_errors():
    from pytorch3d_pointops_amd.functions.chamfer import chamfer_distance

    x = torch.rand(2, 5, 3)
    with pytest.raises(ValueError, match="batch_reduction must be one of"):
        chamfer_distance(x, x, batch_reduction="max")
    with pytest.raises(ValueError, match="point_reduction must be one of"):
        chamfer_distance(x, x, point_reduction="median")
    with pytest.raises(ValueError, match="Batch reduction must be None"):
        chamfer_distance(x, x, point_reduction=None)
    with pytest.raises(ValueError, match="1 or 2 norm"):
        chamfer_distance(x, x, norm=3)
    with pytest.raises(ValueError, match='Features must be None if point_reduction is "max"'):
        chamfer_distance(x, x, point_reduction="max", feature_names=["normals"])
    with pytest.raises(ValueError, match="shape \\(N, P, D\\)"):
        chamfer_distance(torch.rand(5, 3), x)
    with pytest.raises(ValueError, match="shape \\(N,\\)"):
        chamfer_distance(x, x, x_lengths=torch.tensor([5]))
    with pytest.raises(ValueError, match="too long"):
        chamfer_distance(x, x, x_lengths=torch.tensor([5, 6]))
    with pytest.raises(ValueError, match="should be either"):
        chamfer_distance([1, 2], x)


def test_lengths_validation_cache_follows_the_tensor():
    """`lengths_max` remembers `int(lengths.max())` per tensor object and version counter (the validation of the
    reference's chamfer API, functions/chamfer.py:69-70, without a device-to-host copy per call): an in-place write is
    seen, a new tensor is a new entry, and the error text is the reference's."""
    from pytorch3d_pointops_amd.functions._common import lengths_max
    from pytorch3d_pointops_amd.functions.chamfer import _handle_pointcloud_input

    x = torch.rand(2, 5, 3)
    lengths = torch.tensor([5, 4])
    assert lengths_max(lengths) == 5 and lengths_max(lengths) == 5
    _handle_pointcloud_input(x, lengths, None)
    lengths[1] = 6  # in-place: the version counter moves, the remembered value is dropped
    assert lengths_max(lengths) == 6
    with pytest.raises(ValueError, match="too long"):
        _handle_pointcloud_input(x, lengths, None)
    assert lengths_max(torch.tensor([1, 2])) == 2 and lengths_max(torch.zeros(0, dtype=torch.int64)) == 0
    # The documented caveat: a write that bypasses the version counter is not seen (the kernels clamp lengths to the
    # padded size, so a stale maximum can only lose the "too long" error, never read out of bounds) ...
    quiet = torch.tensor([5, 4])
    assert lengths_max(quiet) == 5
    quiet.data[1] = 9
    assert quiet._version == 0 and lengths_max(quiet) == 5
    quiet += 0  # ... until any in-place op on the tensor itself moves the counter
    assert lengths_max(quiet) == 9
    # entries die with their tensors (weakref finalizer), the table does not grow with dead ids
    from pytorch3d_pointops_amd.functions import _common
    import gc

    before = len(_common._MAX_CACHE)
    for _ in range(50):
        lengths_max(torch.tensor([3, 1]))
    gc.collect()
    assert len(_common._MAX_CACHE) <= before + 1


def test_pointclouds_container_against_reference_fixture():
    """The container calls of the reference's examples/pointclouds.py:12-174 and the module-level helpers its callers
    use (update_padded, offset, scale, get_bounding_boxes, join_pointclouds_as_scene, inside_box, extend, split,
    subsample, all_close) on this package's Pointclouds against what the REFERENCE's container returned for the same
    inputs (tests/golden/pointclouds_api.npz, make_golden.py gen_pointclouds)."""
    import numpy as np

    import cases
    from conftest import load_golden
    from pytorch3d_pointops_amd import structures as st

    g = load_golden("pointclouds_api")
    inp = cases.example_clouds()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    pc = st.Pointclouds(points=[T(a) for a in inp["points"]],
                        features={"normals": [T(a) for a in inp["normals"]], "colors": [T(a) for a in inp["colors"]]})
    eq = lambda t, name: np.array_equal(t.numpy(), g[name])  # noqa: E731
    assert eq(pc.points_padded(), "padded") and eq(pc.points_packed(), "packed")
    assert eq(pc.get_features_padded("normals"), "normals_padded") and eq(pc.get_features_packed("colors"), "colors_packed")
    assert eq(pc.cloud_to_packed_first_idx(), "first_idx") and eq(pc.packed_to_cloud_idx(), "packed_to_cloud")
    assert eq(pc.padded_to_packed_idx(), "padded_to_packed")
    assert pc.get_features_list("nope") is None and pc.get_features_packed("nope") is None
    pts, feats = pc.get_cloud(1)
    assert eq(pts, "cloud1_points") and eq(feats["colors"], "cloud1_colors") and sorted(feats) == ["colors", "normals"]
    new_pts = pc.points_padded() * 2.0 + 1.0
    up = pc.update_padded(new_pts)
    assert eq(up.points_packed(), "update_packed") and eq(up.get_features_packed("normals"), "update_normals_packed")
    up2 = pc.update_padded(new_pts, new_features_padded={"normals": pc.get_features_padded("normals") * -1.0})
    assert eq(up2.get_features_list("normals")[1], "update2_normals_list1")
    assert sorted(up2.features_list().keys()) == list(g["update2_names"])
    with pytest.raises(ValueError, match="same number of points"):
        pc.update_padded(new_pts[:, :-1])
    assert eq(st.offset(pc, torch.tensor([0.5, -1.0, 2.0])).points_padded(), "offset_padded")
    sc = st.scale(pc, torch.tensor([2.0, 0.25]))
    assert eq(sc.points_packed(), "scale_packed") and eq(sc.points_padded(), "scale_padded")
    assert eq(pc.points_padded(), "padded")  # (out of place: the original is untouched)
    assert eq(st.get_bounding_boxes(pc), "bboxes")
    scene = st.join_pointclouds_as_scene(pc)
    assert eq(scene.points_padded(), "scene_padded") and eq(scene.get_features_padded("colors"), "scene_colors")
    box = torch.tensor([[[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]], [[0.0, -1.0, -1.0], [2.0, 1.0, 1.0]]])
    assert eq(pc.inside_box(box), "inside_box")
    ext = pc.extend(2)
    assert eq(ext.num_points_per_cloud(), "extend_lengths") and eq(ext.points_padded(), "extend_padded")
    parts = ext.split([1, 3])
    assert eq(parts[1].num_points_per_cloud(), "split1_lengths")
    assert eq(parts[1].get_features_packed("normals"), "split1_normals_packed")
    sub = st.subsample(pc, [100, 5000])
    assert sub.num_points_per_cloud().tolist() == [100, 800] and sub.get_features_list("colors")[0].shape == (100, 3)
    assert st.subsample(pc, 5000) is pc
    assert st.all_close(pc, pc.clone()) and not st.all_close(pc, sc)


def test_graph_capture_has_no_cpu_path():
    """graphs.capture is a GPU tool: CPU tensors raise before anything is captured (no silent eager fallback)."""
    import pytest
    import torch

    from pytorch3d_pointops_amd import graphs

    with pytest.raises(RuntimeError, match="GPU tensors"):
        graphs.capture(lambda p: p * 2, (torch.zeros(2, 8, 3),))
    with pytest.raises(RuntimeError, match="GPU tensors"):
        graphs.capture(lambda: None, ())
