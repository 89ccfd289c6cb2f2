"""Seeded input builders shared by tests/golden/make_golden.py (which runs the
REFERENCE on them in the build container) and by the parity tests (which run the
oracle and the HIP path on the same inputs anywhere).  Inputs are regenerated from
seeds with pytorch3d_pointops_amd.synth, so the fixtures only hold outputs.
"""
import numpy as np

from pytorch3d_pointops_amd import synth


def cloud(seed, shape, scale=1.0):
    return (synth.uniform_f32(seed, shape) * np.float32(scale)).astype(np.float32)


def lattice(seed, N, P, D=3, levels=4):
    """Points on a coarse integer lattice: many exactly tied distances and duplicates
    (the tie regime of SURVEY.md section 3.1; cf. examples/ball_query_on_pointclouds.py:118-125)."""
    return (synth.randint(seed, 0, levels - 1, (N, P, D)).astype(np.float32) * np.float32(0.25)).astype(np.float32)


# ---------------------------------------------------------------- KNN cases
def knn_cases():
    c = {}
    # G1 = BASELINE.json configs[0]: B=2 N=M=1024 K=8 D=3, self-KNN call pattern
    p = cloud(101, (2, 1024, 3))
    c["cfg1_self_k8"] = dict(p1=p, p2=p, l1=np.array([1024, 1024]), l2=np.array([1024, 1024]), K=8, norm=2)
    # G2 ragged: lengths1 != P1, lengths2 < K, K > P2 for one cloud, empty clouds
    c["ragged_k8"] = dict(p1=cloud(102, (4, 200, 3)), p2=cloud(103, (4, 150, 3)),
                          l1=np.array([200, 37, 0, 199]), l2=np.array([150, 5, 77, 0]), K=8, norm=2)
    c["k_gt_p2"] = dict(p1=cloud(104, (2, 50, 3)), p2=cloud(105, (2, 6, 3)),
                        l1=np.array([50, 13]), l2=np.array([6, 3]), K=10, norm=2)
    # G3 ties / duplicates
    lt = lattice(106, 2, 300)
    c["ties_lattice_k16"] = dict(p1=lt, p2=lt, l1=np.array([300, 250]), l2=np.array([300, 123]), K=16, norm=2)
    dup = cloud(107, (1, 64, 3))
    dup = np.concatenate([dup, dup, dup], axis=1)  # every point three times
    c["ties_dup_k5"] = dict(p1=dup[:, :40], p2=dup, l1=np.array([40]), l2=np.array([192]), K=5, norm=2)
    # G4 L1 norm
    c["l1_k4"] = dict(p1=cloud(108, (2, 130, 3)), p2=cloud(109, (2, 257, 3)),
                      l1=np.array([130, 100]), l2=np.array([257, 31]), K=4, norm=1)
    c["l1_ties_k3"] = dict(p1=lattice(110, 1, 90), p2=lattice(111, 1, 140),
                           l1=np.array([90]), l2=np.array([140]), K=3, norm=1)
    # K = 1 (chamfer regime), K = 32 (register cap), K = 40 (generic path), K odd
    c["k1"] = dict(p1=cloud(112, (3, 333, 3)), p2=cloud(113, (3, 1000, 3)),
                   l1=np.array([333, 1, 300]), l2=np.array([1000, 999, 1]), K=1, norm=2)
    c["k32"] = dict(p1=cloud(114, (1, 100, 3)), p2=cloud(115, (1, 700, 3)),
                    l1=np.array([100]), l2=np.array([700]), K=32, norm=2)
    c["k40_generic"] = dict(p1=cloud(116, (2, 60, 3)), p2=cloud(117, (2, 300, 3)),
                            l1=np.array([60, 60]), l2=np.array([300, 20]), K=40, norm=2)
    c["k11"] = dict(p1=cloud(118, (2, 70, 3)), p2=cloud(119, (2, 500, 3)),
                    l1=np.array([70, 70]), l2=np.array([500, 9]), K=11, norm=2)
    # other point dimensions: D=1,2,5,8 (templated) and D=11 (generic)
    for D, seed in ((1, 120), (2, 122), (5, 124), (8, 126), (11, 128)):
        c[f"d{D}_k6"] = dict(p1=cloud(seed, (2, 90, D)), p2=cloud(seed + 1, (2, 210, D)),
                             l1=np.array([90, 45]), l2=np.array([210, 4]), K=6, norm=2)
    # larger coordinates (far from the origin): cancellation regime
    c["offset_k8"] = dict(p1=cloud(130, (1, 256, 3)) + np.float32(1000.0), p2=cloud(131, (1, 512, 3)) + np.float32(1000.0),
                          l1=np.array([256]), l2=np.array([512]), K=8, norm=2)
    # clustered data (non-uniform density)
    cl = (cloud(132, (2, 600, 3)) ** np.float32(4.0)).astype(np.float32)
    c["clustered_k16"] = dict(p1=cl[:, :300], p2=cl, l1=np.array([300, 280]), l2=np.array([600, 555]), K=16, norm=2)
    return c


def knn_backward_cases():
    c = {}
    for name in ("ragged_k8", "ties_lattice_k16", "l1_k4", "k1", "d5_k6", "k_gt_p2"):
        c[name] = knn_cases()[name]
    return c


def grad_for(name, shape):
    seed = 9000 + sum(ord(ch) for ch in name)
    return (synth.uniform_f32(seed, shape) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)


# ---------------------------------------------------------------- ball query
def ball_query_cases():
    c = {}
    p1 = cloud(201, (3, 120, 3))
    p2 = cloud(202, (3, 400, 3))
    for r in (0.1, 0.2, 0.3):
        c[f"ragged_r{r}"] = dict(p1=p1, p2=p2, l1=np.array([120, 33, 0]), l2=np.array([400, 250, 10]), K=16, radius=r)
    c["k500_default"] = dict(p1=cloud(203, (1, 64, 3)), p2=cloud(204, (1, 900, 3)),
                             l1=np.array([64]), l2=np.array([900]), K=500, radius=0.2)
    # strict '<' boundary: lattice spacing 0.25 -> dist2 of neighbours is exactly 0.0625 = 0.25**2
    lt = lattice(205, 2, 200)
    c["boundary_lattice"] = dict(p1=lt, p2=lt, l1=np.array([200, 180]), l2=np.array([200, 77]), K=8, radius=0.25)
    c["d2"] = dict(p1=cloud(206, (2, 80, 2)), p2=cloud(207, (2, 300, 2)),
                   l1=np.array([80, 80]), l2=np.array([300, 150]), K=12, radius=0.15)
    c["d6_generic"] = dict(p1=cloud(208, (1, 50, 6)), p2=cloud(209, (1, 200, 6)),
                           l1=np.array([50]), l2=np.array([200]), K=7, radius=0.6)
    return c


# ---------------------------------------------------------------- FPS
def fps_cases():
    c = {}
    pts = cloud(301, (4, 500, 3))
    c["fixed_k"] = dict(points=pts, lengths=np.array([500, 500, 500, 500]), K=np.array([32] * 4), start=np.zeros(4, np.int64))
    c["per_cloud_k"] = dict(points=pts, lengths=np.array([500, 120, 33, 1]), K=np.array([10, 200, 5, 3]), start=np.zeros(4, np.int64))
    c["start_nonzero"] = dict(points=pts, lengths=np.array([500, 120, 33, 7]), K=np.array([16] * 4), start=np.array([499, 60, 32, 3]))
    dup = np.zeros((1, 40, 3), np.float32) + np.float32(0.5)
    c["all_equal"] = dict(points=dup, lengths=np.array([40]), K=np.array([4]), start=np.zeros(1, np.int64))
    c["lattice_ties"] = dict(points=lattice(302, 2, 300), lengths=np.array([300, 211]), K=np.array([64, 64]), start=np.zeros(2, np.int64))
    c["big_cloud"] = dict(points=cloud(303, (1, 5000, 3)), lengths=np.array([5000]), K=np.array([128]), start=np.zeros(1, np.int64))
    c["d5"] = dict(points=cloud(304, (2, 300, 5)), lengths=np.array([300, 100]), K=np.array([20, 20]), start=np.zeros(2, np.int64))
    return c


# ---------------------------------------------------------------- packed <-> padded
def packed_cases():
    c = {}
    lens = np.array([5, 0, 7, 3, 12])
    c["d3"] = dict(lens=lens, D=3, seed=401, max_size=12)
    c["d1"] = dict(lens=lens, D=1, seed=402, max_size=12)
    c["d4_wide_pad"] = dict(lens=np.array([100, 1, 64]), D=4, seed=403, max_size=128)
    c["d7"] = dict(lens=np.array([33, 2]), D=7, seed=404, max_size=33)
    return c


def packed_inputs(case):
    lens = case["lens"]
    F = int(lens.sum())
    first = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    x = cloud(case["seed"], (F, case["D"]))
    return x, first, F


# ---------------------------------------------------------------- chamfer
def chamfer_inputs(seed=501, N=3, P1=140, P2=170, ragged=True):
    x = cloud(seed, (N, P1, 3))
    y = cloud(seed + 1, (N, P2, 3))
    xn = synth.unit_normals(seed + 2, (N, P1, 3))
    yn = synth.unit_normals(seed + 3, (N, P2, 3))
    if ragged:
        xl = np.array([P1, 50, 99][:N])
        yl = np.array([120, P2, 7][:N])
    else:
        xl = np.array([P1] * N)
        yl = np.array([P2] * N)
    w = np.array([1.0, 0.5, 2.0][:N], np.float32)
    return dict(x=x, y=y, xn=xn, yn=yn, xl=xl, yl=yl, w=w)


def chamfer_variants():
    v = []
    for pr in ("mean", "sum", "max", None):
        for br in ("mean", "sum", None):
            if pr is None and br is not None:
                continue
            for single in (False, True):
                for use_w in (False, True):
                    for feats in (False, True):
                        if pr == "max" and feats:
                            continue
                        v.append(dict(point_reduction=pr, batch_reduction=br, single_directional=single,
                                      use_weights=use_w, features=feats, abs_cosine=True, norm=2))
    v.append(dict(point_reduction="mean", batch_reduction="mean", single_directional=False,
                  use_weights=False, features=True, abs_cosine=False, norm=2))
    v.append(dict(point_reduction="mean", batch_reduction="mean", single_directional=False,
                  use_weights=True, features=False, abs_cosine=True, norm=1))
    v.append(dict(point_reduction="sum", batch_reduction=None, single_directional=True,
                  use_weights=False, features=True, abs_cosine=False, norm=1))
    return v


def variant_key(v):
    return "pr={point_reduction}|br={batch_reduction}|sd={single_directional}|w={use_weights}|f={features}|abs={abs_cosine}|n={norm}".format(**v)


# ---------------------------------------------------------------- sample_pdf
def sample_pdf_cases():
    """bins sorted in [0,1), non-negative weights (some empty bins), quantiles u incl. 0 and 1.
    Batch sizes avoid 5: the reference's CPU thread partition (sample_pdf_cpu.cpp:120-141) hands
    thread 3 the rows [4,6) of a 5-row batch and writes out of bounds."""
    c = {}
    for name, (batch, nb, ns, seed) in {"b8_64x128": (8, 64, 128, 701), "b12_7x33": (12, 7, 33, 702),
                                        "b1_1x5": (1, 1, 5, 703), "b4_128x64": (4, 128, 64, 704)}.items():
        bins = np.sort(synth.uniform_f32(seed, (batch, nb + 1)), axis=1).astype(np.float32)
        w = synth.uniform_f32(seed + 10, (batch, nb))
        w[0, : nb // 3] = 0.0
        u = synth.uniform_f32(seed + 20, (batch, ns))
        u[0, 0] = 0.0
        u[0, -1] = 1.0
        c[name] = dict(bins=bins, weights=w, u=u, eps=1e-5)
    return c


# ---------------------------------------------------------------- the reference's example script
def example_clouds():
    """Clouds shaped like examples/knn_on_pointclouds.py:15-58 of the reference: a 1500-point sphere shell
    (radius 0.2..1) and an 800-point ellipsoid shell (radius 0.4..1, x stretched 1.5, y squeezed 0.8), unit
    normals and random colours.  (make_golden.py stores these arrays in the fixture; tests read them back.)"""
    def shell(seed, n, r0, r1, sx, sy):
        u = synth.uniform_f32(seed, (n, 3)).astype(np.float64)
        theta, phi, r = u[:, 0] * 2 * np.pi, u[:, 1] * np.pi, u[:, 2] * (r1 - r0) + r0
        return np.stack([r * np.sin(phi) * np.cos(theta) * sx, r * np.sin(phi) * np.sin(theta) * sy,
                         r * np.cos(phi)], axis=1).astype(np.float32)

    def unit(a):
        a = a.astype(np.float64)
        return (a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)).astype(np.float32)

    p0, p1 = shell(901, 1500, 0.2, 1.0, 1.0, 1.0), shell(902, 800, 0.4, 1.0, 1.5, 0.8)
    n1 = unit(p1 / np.array([2.25, 0.64, 1.0]))
    return dict(points=[p0, p1], normals=[unit(p0), n1],
                colors=[synth.uniform_f32(903, (1500, 3)), synth.uniform_f32(904, (800, 3))])


def example2_inputs():
    """Inputs shaped like the reference's other self-checking example scripts (ball_query_on_pointclouds.py:20-47,
    fps_on_pointclouds.py:22-64, chamfer_loss.py:17-28, packed_to_padded_on_pointclouds.py:22-64,
    utils_on_pointclouds.py:23-66): the scripts draw from torch.rand / randn; here the same shapes and value ranges come
    from the seeded generator (make_golden.py stores nothing but outputs; these arrays are regenerated by the tests)."""
    def u(seed, shape):
        return synth.uniform_f32(seed, shape).astype(np.float64)

    def shell(seed, n, r0, r1, s=(1.0, 1.0, 1.0)):
        a = u(seed, (n, 3))
        theta, phi, r = a[:, 0] * 2 * np.pi, a[:, 1] * np.pi, a[:, 2] * (r1 - r0) + r0
        return np.stack([r * np.sin(phi) * np.cos(theta) * s[0], r * np.sin(phi) * np.sin(theta) * s[1],
                         r * np.cos(phi) * s[2]], axis=1).astype(np.float32)

    def normalish(seed, shape, scale=1.0):  # sum of 4 uniforms, centred: bell-shaped like randn, fully reproducible
        a = u(seed, shape + (4,)).sum(-1) - 2.0
        return (a * (scale * 1.7320508)).astype(np.float32)

    def cylinder(seed, n, rmax):
        a = u(seed, (n, 3))
        theta, r, z = a[:, 0] * 2 * np.pi, a[:, 1] * rmax, a[:, 2] * 2 - 1
        return np.stack([r * np.cos(theta), r * np.sin(theta), z], axis=1).astype(np.float32)

    lin = np.linspace(-1, 1, 10, dtype=np.float32)
    gx, gy, gz = np.meshgrid(lin, lin, lin, indexing="ij")
    th = np.linspace(0, 2 * np.pi, 150, dtype=np.float32)
    circles = np.stack([np.concatenate([0.8 * np.cos(th), 0.4 * np.cos(th)]),
                        np.concatenate([0.8 * np.sin(th), 0.4 * np.sin(th)])], axis=1).astype(np.float32)
    sizes = [50, 100, 5000, 10, 1000, 20, 3000]
    return dict(
        ball=dict(points=[shell(3101, 2000, 0.5, 1.0), (u(3102, (500, 3)) * 4 - 2).astype(np.float32)],
                  lattice=np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1).astype(np.float32)),
        fps=dict(points=[shell(3111, 1000, 0.5, 1.0), (u(3112, (800, 3)) * 2 - 1).astype(np.float32),
                         cylinder(3113, 1200, 1.0)],
                 colors0=synth.uniform_f32(3114, (1000, 3)), single=(u(3115, (500, 3)) * 2 - 1).astype(np.float32),
                 compare=synth.uniform_f32(3116, (1, 2000, 3)), circles=circles),
        chamfer=dict(p1=normalish(3121, (2, 100, 3)), p2=normalish(3122, (2, 120, 3)),
                     n1=normalish(3123, (2, 100, 3)), n2=normalish(3124, (2, 120, 3)),
                     c1=normalish(3125, (2, 100, 3)), c2=normalish(3126, (2, 120, 3))),
        packed=dict(points=[normalish(3131, (100, 3), 0.5), cylinder(3132, 500, 2.0), normalish(3133, (2000, 3), 1.5),
                            normalish(3134, (25, 3), 0.2)],
                    intensities=[synth.uniform_f32(3135 + i, (n, 1)) for i, n in enumerate((100, 500, 2000, 25))],
                    var_points=[normalish(3141 + i, (n, 3)) for i, n in enumerate(sizes)],
                    var_features=[normalish(3151 + i, (n, 16)) for i, n in enumerate(sizes)]),
        utils=dict(points=[shell(3161, 1000, 0.2, 1.0), shell(3162, 800, 0.4, 1.0, (4.0, 0.5, 1.0))],
                   weights=[(u(3163, (1000, 1)) * 0.5 + 0.5).astype(np.float32),
                            (u(3164, (800, 1)) * 0.8 + 0.2).astype(np.float32)],
                   values=[normalish(3165, (1000, 4)), normalish(3166, (800, 4))]),
    )
