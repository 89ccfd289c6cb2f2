#!/bin/bash
# usage: bash tools/prof_quick.sh <tag> [POINTOPS_DEBUG value]   -> gpurun_out/<tag>_kernel_stats.csv (+ bench line)
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
export POINTOPS_DEBUG="$2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG.trace -- python $R/bench.py --steps 10 --warmup 3 --cpu-sample-queries 0 > $R/gpurun_out/$TAG.bench.json 2> /dev/null
cp $(find $R/gpurun_out/$TAG.trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/$TAG.trace
cut -d, -f1-4 $R/gpurun_out/${TAG}_kernel_stats.csv | sed 's/void pointops:://' | cut -c1-110 | head -${3:-8}
