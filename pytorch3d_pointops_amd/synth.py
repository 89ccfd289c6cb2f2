"""Deterministic synthetic point clouds (no torch RNG, no reference needed).

splitmix64 counter stream -> top 24 bits -> fp32 in [0, 1).  The same seed gives
bit-identical inputs in the build container and on the GPU box, which is what lets
tests/golden digests and bench.py workloads be regenerated anywhere
(SURVEY.md section 8d "Synthetic inputs").
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_f32(seed: int, shape) -> np.ndarray:
    """fp32 uniform in [0,1) with 24 random mantissa bits."""
    n = int(np.prod(shape))
    bits = (splitmix64(seed, n) >> np.uint64(40)).astype(np.float32)
    return (bits * np.float32(2.0 ** -24)).reshape(shape)


def randint(seed: int, lo: int, hi: int, shape) -> np.ndarray:
    """int64 uniform in [lo, hi] inclusive."""
    n = int(np.prod(shape))
    r = splitmix64(seed, n) >> np.uint64(11)
    return (lo + (r % np.uint64(hi - lo + 1)).astype(np.int64)).reshape(shape)


def unit_normals(seed: int, shape) -> np.ndarray:
    """fp32 unit vectors (last dim = 3) from the same stream."""
    v = uniform_f32(seed, shape) * np.float32(2.0) - np.float32(1.0)
    nrm = np.sqrt((v * v).sum(-1, keepdims=True)).astype(np.float32)
    return (v / np.maximum(nrm, np.float32(1e-6))).astype(np.float32)


# ------------------------------------------------------------------ non-uniform clouds (benchmarks and tests)
DISTRIBUTIONS = ("uniform", "sphere", "planes", "clustered_u4", "half_in_cluster", "aniso_100")


def distribution(name: str, seed: int, n: int) -> np.ndarray:
    """(n, 3) fp32 cloud of a named shape, all inside about [0,1]^3:
    uniform           the cube;
    sphere            points ON a sphere surface (a 2-D manifold: most grid cells are empty);
    planes            two parallel planes z = 0.3 / 0.7 with 1e-3 noise (scanned surfaces);
    clustered_u4      u^4 per coordinate: density rising steeply towards the origin corner;
    half_in_cluster   half of the points inside a 1e-3 cube, the rest uniform (one cell holds half the cloud);
    aniso_100         a 1 x 1 x 0.01 slab (100:1 anisotropic bounding box)."""
    u = uniform_f32(seed, (n, 3)).astype(np.float64)
    if name == "uniform":
        p = u
    elif name == "sphere":
        z = 2.0 * u[:, 0] - 1.0
        phi = 2.0 * np.pi * u[:, 1]
        r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        p = 0.5 + 0.5 * np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)
    elif name == "planes":
        p = u.copy()
        p[:, 2] = np.where(np.arange(n) % 2 == 0, 0.3, 0.7) + (u[:, 2] - 0.5) * 2e-3
    elif name == "clustered_u4":
        p = u ** 4
    elif name == "half_in_cluster":
        p = u.copy()
        p[::2] = 0.5 + (u[::2] - 0.5) * 1e-3
    elif name == "aniso_100":
        p = u.copy()
        p[:, 2] *= 0.01
    else:
        raise ValueError(name)
    return p.astype(np.float32)
