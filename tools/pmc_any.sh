#!/bin/bash
# usage: bash tools/pmc_any.sh <tag> "<counters>" [POINTOPS_DEBUG]  -> per-kernel sums of the counters (per launch) for the lane kernel
TAG=$1; CNT=$2
R=${GRAFT_REPO_ROOT:-$PWD}
export POINTOPS_DEBUG="$3"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/$TAG.pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-queries 0 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/$TAG.pmc/*/*counter_collection.csv")[0]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][-40:]
    tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,c in tot.items():
    if "lane" in k or "quad" in k: print(k, {a: round(v/len(n[k])) for a,v in c.items()})
PY
rm -rf $R/gpurun_out/$TAG.pmc
