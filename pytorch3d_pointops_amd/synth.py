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
