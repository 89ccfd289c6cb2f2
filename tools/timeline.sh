#!/bin/bash
# usage: bash tools/timeline.sh <POINTOPS_DEBUG> : start/end (us, relative) of the kernels of the last bench step
R=${GRAFT_REPO_ROOT:-$PWD}
export POINTOPS_DEBUG="$1"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 3 --warmup 2 --cpu-sample-queries 0 --reuse-steps 0 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/tl/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "pointops" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: find last grid_bbox occurrences
starts=[i for i,r in enumerate(rows) if "grid_bbox" in r["Kernel_Name"]]
import os
np_=int(os.environ.get("NP","1"))
i0=starts[-np_]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    print("%8.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3, r.get("Queue_Id","?"), r["Kernel_Name"].replace("void pointops::","")[:48]))
PY
