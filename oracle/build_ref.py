#!/usr/bin/env python3
"""Build recipe for ``oracle/_ref/_C*.so`` -- TEST INFRASTRUCTURE ONLY.

Compiles the reference's own CPU translation units *where they lie* under
``/root/reference/pytorch3d_pointops/csrc`` (``ext.cpp`` plus the five
``*_cpu.cpp`` files; no CUDA, no ``WITH_CUDA``) with plain ``g++`` against the
libtorch headers that ship with the image's PyTorch, into
``oracle/_ref/_C<EXT_SUFFIX>``.  It does not run the reference's ``setup.py``
and copies no reference source into this repository; the output directory is
git-ignored (``oracle/_ref/``) but travels to the GPU box with the snapshot, so
the compiled reference can serve there as a checker / CPU baseline.

Flags mirror what the reference's CppExtension build used:
``-O2 -std=c++17`` and *no* ``-mfma`` / ``-march=native``, so ``dist += diff*diff``
stays an unfused multiply + add (the parity rule of SURVEY.md §3.1).

If ``/root/reference`` is absent (GPU box) this is a no-op: the prebuilt file
that travelled with the snapshot is used as is.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = "/root/reference/pytorch3d_pointops/csrc"
OUT_DIR = os.path.join(HERE, "_ref")
SOURCES = [
    "ext.cpp",
    "knn/knn_cpu.cpp",
    "ball_query/ball_query_cpu.cpp",
    "sample_farthest_points/sample_farthest_points_cpu.cpp",
    "packed_to_padded_tensor/packed_to_padded_tensor_cpu.cpp",
    "sample_pdf/sample_pdf_cpu.cpp",  # ext.cpp binds it; unused by the hot path
]


def ref_so_path() -> str:
    return os.path.join(OUT_DIR, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force: bool = False, verbose: bool = True) -> str:
    out = ref_so_path()
    if not os.path.isdir(REF_CSRC):
        if verbose:
            print(f"[oracle/_ref] {REF_CSRC} not present; using prebuilt {out} if any")
        return out
    srcs = [os.path.join(REF_CSRC, s) for s in SOURCES]
    if (
        not force
        and os.path.exists(out)
        and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs)
        and os.path.getmtime(out) >= os.path.getmtime(__file__)
    ):
        return out
    from torch.utils.cpp_extension import include_paths, library_paths
    import torch

    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    common = ["g++", "-O2", "-std=c++17", "-fPIC", "-w",
              "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H",
              f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
              f"-I{REF_CSRC}", f"-I{sysconfig.get_paths()['include']}"]
    common += [f"-I{p}" for p in include_paths()]
    procs = []
    for s in srcs:
        o = os.path.join(OUT_DIR, os.path.basename(s).replace(".cpp", ".o"))
        objs.append(o)
        procs.append(subprocess.Popen(common + ["-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("oracle/_ref: compiling a reference TU failed")
    link = ["g++", "-shared", "-o", out] + objs
    for lp in library_paths():
        link += [f"-L{lp}", f"-Wl,-rpath,{lp}"]
    link += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python"]
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)
    if verbose:
        print(f"[oracle/_ref] built {out}")
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
