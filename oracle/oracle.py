"""ctypes/numpy front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module (see oracle/pointops_oracle.c header).  It is the checker,
never the product path.

Two back-ends with one calling convention (numpy in, numpy out):

* ``Oracle()``      -- liboracle.so, the plain-C restatement (always available
                       after ``make -C oracle liboracle.so``); ``kind == "port"``.
* ``RefOracle()``   -- oracle/_ref/_C*.so, the reference's own CPU kernels compiled
                       by oracle/build_ref.py; ``kind == "reference"``.  Returns
                       None from ``load_ref()`` when the file is absent.

Function names and argument order follow the reference's ``_C`` operator boundary
(/root/reference/pytorch3d_pointops/csrc/ext.cpp:16-26): note that
``knn_points_idx`` and ``ball_query`` return ``(idx, dists)``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i64 = ctypes.c_int64


def build(verbose: bool = False) -> str:
    """Compile liboracle.so with gcc if missing or stale."""
    src = os.path.join(_HERE, "pointops_oracle.c")
    if (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
               "-shared", "-o", _LIB_PATH, src, "-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return _LIB_PATH


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64a(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _pf(a):
    return a.ctypes.data_as(_f32p)


def _pi(a):
    return a.ctypes.data_as(_i64p)


class Oracle:
    kind = "port"

    def __init__(self):
        build()
        self.lib = ctypes.CDLL(_LIB_PATH)
        L = self.lib
        L.oracle_knn_points_idx.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64, _i64, _i64, _i64,
                                            ctypes.c_int, _i64, _i64p, _f32p]
        L.oracle_knn_points_backward.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64p, _f32p,
                                                 _i64, _i64, _i64, _i64, _i64, ctypes.c_int,
                                                 _f32p, _f32p]
        L.oracle_ball_query.argtypes = [_f32p, _f32p, _i64p, _i64p, _i64, _i64, _i64, _i64, _i64,
                                        ctypes.c_float, _i64p, _f32p]
        L.oracle_sample_farthest_points.argtypes = [_f32p, _i64p, _i64p, _i64p, _i64, _i64, _i64,
                                                    _i64, _i64p]
        L.oracle_sample_pdf.argtypes = [_f32p, _f32p, _f32p, _i64, _i64, _i64, ctypes.c_float]
        L.oracle_sample_pdf.restype = None
        L.oracle_packed_to_padded.argtypes = [_f32p, _i64p, _i64, _i64, _i64, _i64, _f32p]
        L.oracle_padded_to_packed.argtypes = [_f32p, _i64p, _i64, _i64, _i64, _i64, _f32p]
        for f in (L.oracle_knn_points_idx, L.oracle_knn_points_backward, L.oracle_ball_query,
                  L.oracle_sample_farthest_points, L.oracle_packed_to_padded,
                  L.oracle_padded_to_packed):
            f.restype = None

    # -- KNN ---------------------------------------------------------------
    def knn_points_idx(self, p1, p2, lengths1, lengths2, norm, K, version=-1):
        p1, p2 = _f32(p1), _f32(p2)
        l1, l2 = _i64a(lengths1), _i64a(lengths2)
        N, P1, D = p1.shape
        P2 = p2.shape[1]
        idx = np.empty((N, P1, K), np.int64)
        dists = np.empty((N, P1, K), np.float32)
        self.lib.oracle_knn_points_idx(_pf(p1), _pf(p2), _pi(l1), _pi(l2), N, P1, P2, D,
                                       int(norm), K, _pi(idx), _pf(dists))
        return idx, dists

    def knn_points_backward(self, p1, p2, lengths1, lengths2, idxs, norm, grad_dists):
        p1, p2, g = _f32(p1), _f32(p2), _f32(grad_dists)
        l1, l2, idxs = _i64a(lengths1), _i64a(lengths2), _i64a(idxs)
        N, P1, D = p1.shape
        P2 = p2.shape[1]
        K = idxs.shape[2]
        g1 = np.empty((N, P1, D), np.float32)
        g2 = np.empty((N, P2, D), np.float32)
        self.lib.oracle_knn_points_backward(_pf(p1), _pf(p2), _pi(l1), _pi(l2), _pi(idxs), _pf(g),
                                            N, P1, P2, D, K, int(norm), _pf(g1), _pf(g2))
        return g1, g2

    # -- ball query ----------------------------------------------------------
    def ball_query(self, p1, p2, lengths1, lengths2, K, radius):
        p1, p2 = _f32(p1), _f32(p2)
        l1, l2 = _i64a(lengths1), _i64a(lengths2)
        N, P1, D = p1.shape
        P2 = p2.shape[1]
        idx = np.empty((N, P1, K), np.int64)
        dists = np.empty((N, P1, K), np.float32)
        self.lib.oracle_ball_query(_pf(p1), _pf(p2), _pi(l1), _pi(l2), N, P1, P2, D, K,
                                   float(radius), _pi(idx), _pf(dists))
        return idx, dists

    # -- FPS -----------------------------------------------------------------
    def sample_farthest_points(self, points, lengths, K, start_idxs):
        points = _f32(points)
        lengths, K, start_idxs = _i64a(lengths), _i64a(K), _i64a(start_idxs)
        N, P, D = points.shape
        max_K = int(K.max()) if K.size else 0
        out = np.empty((N, max_K), np.int64)
        self.lib.oracle_sample_farthest_points(_pf(points), _pi(lengths), _pi(K), _pi(start_idxs),
                                               N, P, D, max_K, _pi(out))
        return out

    # -- sample_pdf: returns the samples for quantiles u (the reference works in place) --
    def sample_pdf(self, bins, weights, u, eps):
        bins, weights = _f32(bins), _f32(weights)
        out = np.array(u, dtype=np.float32, copy=True, order="C")
        B, nb1 = bins.shape
        self.lib.oracle_sample_pdf(_pf(bins), _pf(weights), _pf(out), B, nb1 - 1, out.shape[1], float(eps))
        return out

    # -- packed <-> padded ---------------------------------------------------
    def packed_to_padded(self, inputs_packed, first_idxs, max_size):
        x = _f32(inputs_packed)
        f = _i64a(first_idxs)
        F, D = x.shape
        B = f.shape[0]
        out = np.empty((B, max_size, D), np.float32)
        self.lib.oracle_packed_to_padded(_pf(x), _pi(f), F, B, max_size, D, _pf(out))
        return out

    def padded_to_packed(self, inputs_padded, first_idxs, num_inputs):
        x = _f32(inputs_padded)
        f = _i64a(first_idxs)
        B, M, D = x.shape
        out = np.empty((num_inputs, D), np.float32)
        self.lib.oracle_padded_to_packed(_pf(x), _pi(f), num_inputs, B, M, D, _pf(out))
        return out


class RefOracle:
    """The reference's own compiled CPU kernels (oracle/_ref), same interface."""

    kind = "reference"

    def __init__(self, module):
        self.m = module

    @staticmethod
    def _t(a, dtype):
        import torch

        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype))

    def knn_points_idx(self, p1, p2, lengths1, lengths2, norm, K, version=-1):
        i, d = self.m.knn_points_idx(self._t(p1, np.float32), self._t(p2, np.float32),
                                     self._t(lengths1, np.int64), self._t(lengths2, np.int64),
                                     int(norm), int(K), int(version))
        return i.numpy(), d.numpy()

    def knn_points_backward(self, p1, p2, lengths1, lengths2, idxs, norm, grad_dists):
        a, b = self.m.knn_points_backward(self._t(p1, np.float32), self._t(p2, np.float32),
                                          self._t(lengths1, np.int64), self._t(lengths2, np.int64),
                                          self._t(idxs, np.int64), int(norm),
                                          self._t(grad_dists, np.float32))
        return a.numpy(), b.numpy()

    def ball_query(self, p1, p2, lengths1, lengths2, K, radius):
        i, d = self.m.ball_query(self._t(p1, np.float32), self._t(p2, np.float32),
                                 self._t(lengths1, np.int64), self._t(lengths2, np.int64),
                                 int(K), float(radius))
        return i.numpy(), d.numpy()

    def sample_farthest_points(self, points, lengths, K, start_idxs):
        return self.m.sample_farthest_points(self._t(points, np.float32), self._t(lengths, np.int64),
                                             self._t(K, np.int64),
                                             self._t(start_idxs, np.int64)).numpy()

    def sample_pdf(self, bins, weights, u, eps):
        out = self._t(np.array(u, dtype=np.float32, copy=True), np.float32)
        self.m.sample_pdf(self._t(bins, np.float32), self._t(weights, np.float32), out, float(eps))
        return out.numpy()

    def packed_to_padded(self, inputs_packed, first_idxs, max_size):
        return self.m.packed_to_padded(self._t(inputs_packed, np.float32),
                                       self._t(first_idxs, np.int64), int(max_size)).numpy()

    def padded_to_packed(self, inputs_padded, first_idxs, num_inputs):
        return self.m.padded_to_packed(self._t(inputs_padded, np.float32),
                                       self._t(first_idxs, np.int64), int(num_inputs)).numpy()


def load_ref():
    """Load oracle/_ref/_C*.so (built by oracle/build_ref.py) or return None."""
    import importlib.util

    from .build_ref import ref_so_path

    path = ref_so_path()
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    spec = importlib.util.spec_from_file_location("_C", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return RefOracle(mod)


# ---------------------------------------------------------------------------
# Python-level restatements of the reference wrappers that sit between the
# kernels and the user (numpy; small sizes).  Used to check knn_gather /
# masked_gather / chamfer on the GPU box where the reference cannot travel.
# ---------------------------------------------------------------------------
def knn_gather(x, idx, lengths=None):
    """fn/knn.py:200-250: x_out[n,l,k] = x[n, idx[n,l,k]], zero where k >= lengths[n]."""
    x = np.asarray(x)
    idx = np.asarray(idx)
    N, M, U = x.shape
    _, L, K = idx.shape
    out = x[np.arange(N)[:, None, None], idx]  # (N, L, K, U)
    if lengths is not None:
        lengths = np.asarray(lengths)
        mask = lengths[:, None] <= np.arange(K)[None]
        out = out.copy()
        out[np.broadcast_to(mask[:, None, :], (N, L, K))] = 0.0
    return out


def masked_gather(points, idx):
    """fn/utils.py:20-65: gather with -1 -> 0 padding."""
    points = np.asarray(points)
    idx = np.asarray(idx)
    N = points.shape[0]
    m = idx == -1
    safe = np.where(m, 0, idx)
    if idx.ndim == 3:
        out = points[np.arange(N)[:, None, None], safe]
    elif idx.ndim == 2:
        out = points[np.arange(N)[:, None], safe]
    else:
        raise ValueError("idx format is not supported %s" % repr(idx.shape))
    out = out.copy()
    out[m] = 0.0
    return out
