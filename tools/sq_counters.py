#!/usr/bin/env python3
"""Per-kernel SQ counters of one rocprofv3 --pmc pass of `bench.py` -> JSON.

  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
            SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d DIR -- python bench.py ...
  python tools/sq_counters.py DIR <launches incl. warmup> out.json

SQ_INSTS_* count wave instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 =
shader cycles) summed over waves (MI355X_MICROARCH.md, cycle constants).  Per launch of knn_points_idx (all
kernels of the op): executed VALU instructions -> the `roofline.valu_issue` block of bench.py;
valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the dominant kernel = share of its wave-cycles spent
issuing VALU.  The JSON records the digest of the kernel sources it was measured on.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    d, launches, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:90]
        if "pointops::" not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    kernels = {}
    for k, c in tot.items():
        kernels[k] = {n: v / launches for n, v in sorted(c.items())}
        kernels[k]["duration_us_per_launch_under_pmc"] = dur[k] / launches / 1e3
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            kernels[k]["valu_busy_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
            kernels[k]["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / wc
            kernels[k]["wait_inst_any_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
    dom = max(kernels, key=lambda k: kernels[k].get("SQ_WAVE_CYCLES", 0.0))
    import bench  # kernel_source_digest (repo root on sys.path)

    res = {
        "valu_insts_per_launch": sum(v.get("SQ_INSTS_VALU", 0.0) for v in kernels.values()),
        "dominant_kernel": dom,
        "valu_busy_frac_dominant_kernel": kernels[dom].get("valu_busy_frac"),
        "launches_profiled": launches,
        "kernel_source_digest": bench.kernel_source_digest(),
        "units": "SQ_INSTS_*: wave instructions; *_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*: quad-cycles summed over waves",
        "kernels": kernels,
    }
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: res[k] for k in ("valu_insts_per_launch", "dominant_kernel", "valu_busy_frac_dominant_kernel")}))


if __name__ == "__main__":
    main()
