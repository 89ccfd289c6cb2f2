"""Build a tuning variant of libpointops_amd.so: one translation unit recompiled with extra -D flags,
the other objects reused from the regular build.  Select it at run time with POINTOPS_AMD_LIB=<path>.

  python tools/build_variant.py g4 knn_grid_d3.hip -DPOINTOPS_LANE_GROUP=4
  -> pytorch3d_pointops_amd/lib/variants/libpointops_amd_lpq4.so
"""
import glob
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch3d_pointops_amd import build as B  # noqa: E402


def main():
    name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()
    out_dir = os.path.join(B.LIB_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(os.path.join(B.OBJ_DIR, "variants"), exist_ok=True)  # (build/variants/ is .gpurunignore'd)
    units = unit.split(",")  # (templates without the run-word parameter are instantiated in knn_grid_dN.hip AND knn_grid_dNw.hip:
    new_objs = []            # the linker keeps one copy, so a flag that changes them must reach both units)
    procs = []
    for u in units:
        obj = os.path.join(B.OBJ_DIR, "variants", f"{u}.{name}.o")
        procs.append(subprocess.Popen([B.HIPCC] + B.CXXFLAGS + flags + ["-c", os.path.join(B.CSRC, u), "-o", obj]))
        new_objs.append(obj)
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("compile failed")
    skip = {u + ".o" for u in units}
    objs = [o for o in sorted(glob.glob(os.path.join(B.OBJ_DIR, "*.hip.o"))) if os.path.basename(o) not in skip]
    lib = os.path.join(out_dir, f"libpointops_amd_{name}.so")
    subprocess.check_call([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib] + objs + new_objs)
    print(lib)


if __name__ == "__main__":
    main()
