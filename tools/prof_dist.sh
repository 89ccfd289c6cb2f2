#!/bin/bash
# usage: bash tools/prof_dist.sh <dist> [POINTOPS_DEBUG] : kernel stats of knn_points (B=32, N=65536, K=16) on one distribution
R=${GRAFT_REPO_ROOT:-$PWD}
export POINTOPS_DEBUG="$2"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -- python $R/tools/run_dist.py $1 32 65536 16 5 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/pd/*/*kernel_stats.csv")[0]
tot=0
rows=list(csv.DictReader(open(f)))
for r in rows[:12]:
    print("  %-64s %5s %9.1f us" % (r["Name"].replace("void pointops::","")[:64], r["Calls"], float(r["AverageNs"])/1e3))
print("  sum per call: %.1f us" % (sum(float(r["TotalDurationNs"]) for r in rows if "pointops" in r["Name"])/5/1e3))
PY
