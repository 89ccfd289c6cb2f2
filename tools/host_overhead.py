#!/usr/bin/env python3
"""Host-side cost of one small knn_points call (BASELINE.json configs[0]: B=2, N=M=1024, K=8): the same call through
the public function, through _C, and through the bare C ABI with preallocated outputs; back to back and one at a time
(synchronised after every call)."""
import cProfile, pstats, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch3d_pointops_amd import synth, _C
from pytorch3d_pointops_amd.functions import knn_points
dev = torch.device("cuda:0")
a = torch.from_numpy(synth.uniform_f32(1, (2, 1024, 3))).to(dev)
L = torch.full((2,), 1024, dtype=torch.int64, device=dev)
idx = torch.empty((2, 1024, 8), dtype=torch.int64, device=dev)
d = torch.empty((2, 1024, 8), dtype=torch.float32, device=dev)
lib = _C._lib
wsb = lib.pointops_knn_workspace_bytes(2, 1024, 1024, 3, 8, -1)
ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
st = _C._stream()


def raw():
    lib.pointops_knn_points_idx_reuse(a.data_ptr(), a.data_ptr(), L.data_ptr(), L.data_ptr(), 2, 1024, 1024, 3, 2, 8, -1,
                                      idx.data_ptr(), d.data_ptr(), ws.data_ptr(), wsb, 0, st)


ap, Lp, ip, dp, wp = a.data_ptr(), L.data_ptr(), idx.data_ptr(), d.data_ptr(), ws.data_ptr()


def raw_noptr():
    lib.pointops_knn_points_idx_reuse(ap, ap, Lp, Lp, 2, 1024, 1024, 3, 2, 8, -1, ip, dp, wp, wsb, 0, st)


def timed(f, n=3000, sync=False):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if sync:
        for _ in range(n):
            f(); torch.cuda.synchronize()
    else:
        for _ in range(n):
            f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


rows = [("C ABI, pointers precomputed", raw_noptr), ("C ABI via ctypes + data_ptr()", raw),
        ("_C.knn_points_idx", lambda: _C.knn_points_idx(a, a, L, L, 2, 8, -1)),
        ("functions.knn_points", lambda: knn_points(a, a, K=8)),
        ("functions.knn_points(lengths)", lambda: knn_points(a, a, L, L, K=8)),
        ("torch.empty x2 only", lambda: (torch.empty((2, 1024, 8), dtype=torch.int64, device=dev),
                                         torch.empty((2, 1024, 8), dtype=torch.float32, device=dev)))]
for name, f in rows:
    print("%-34s back to back %6.1f us   one at a time %6.1f us" % (name, timed(f), timed(f, sync=True)), flush=True)
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000):
        knn_points(a, a, K=8)
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
