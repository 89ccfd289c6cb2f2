#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into HBM-side bytes
per bench step, following /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
  * FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC slots);
  * both counters are in KiB;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads,
    so the read side is doubled; WRITE_SIZE is exact for streaming stores.
usage: pmc_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <steps incl. warmup> <out.json>
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"].split("(")[0][:80]] += float(r["Counter_Value"])
    return tot


def main():
    fdir, wdir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe = per_kernel(fdir, "FETCH_SIZE")
    wr = per_kernel(wdir, "WRITE_SIZE")
    keep = lambda k: "pointops::" in k or "rocclr_fill" in k
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if keep(k):
            kernels[k] = {"fetch_bytes_per_step": 2.0 * fe.get(k, 0.0) * 1024 / steps,
                          "write_bytes_per_step": wr.get(k, 0.0) * 1024 / steps}
    total = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in kernels.values())
    import bench  # kernel_source_digest: ties the numbers to the code they were measured on

    json.dump({"traffic_bytes_per_step": total, "steps_profiled": steps, "kernel_source_digest": bench.kernel_source_digest(),
               "correction": "FETCH_SIZE KiB x2 (gfx950 wide-read undercount) + WRITE_SIZE KiB, separate --pmc passes",
               "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps({"traffic_bytes_per_step": total}))


if __name__ == "__main__":
    main()
