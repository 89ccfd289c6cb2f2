"""In-tree build of libpointops_amd.so with hipcc for gfx950 (no torch headers).

    python -m pytorch3d_pointops_amd.build [--force] [--verbose]

Every ``csrc/*.hip`` is compiled to an object (in parallel) and linked into
``pytorch3d_pointops_amd/lib/libpointops_amd.so``.  The .so is git-ignored but
travels to the GPU box with the repo snapshot.  hipcc cross-compiles without a GPU.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpointops_amd.so")
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
    # parity rule: unfused fp32; also set per-TU via pragma
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]


def _newest_dep_mtime():
    deps = glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(HERE, "..", "include", "pointops_amd.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps)


def _compile(src, obj, verbose):
    cmd = [HIPCC] + CXXFLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources found under " + CSRC)
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    dep_m = _newest_dep_mtime()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), dep_m):
            jobs.append((s, o))
    if jobs:
        if not os.path.exists(HIPCC):
            raise RuntimeError(f"{HIPCC} not found and objects are stale: cannot build libpointops_amd.so")
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], verbose), jobs))
    if jobs or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(p)
