#!/bin/bash
# usage (GPU box, repo root): bash tools/quad_variants.sh <variant>...  -- per variant ("main" = the regular library) the average
# duration of every knn_grid_* kernel of the headline call (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/quad_variants
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "main" ]; then unset POINTOPS_AMD_LIB; else export POINTOPS_AMD_LIB=$R/pytorch3d_pointops_amd/lib/variants/libpointops_amd_$v.so; fi
  rm -rf $OUT/$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python $R/bench.py --steps 10 --warmup 3 --reuse-steps 0 --cpu-sample-queries 0 > $OUT/$v.json 2> /dev/null
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(python -c "import json;print(json.load(open('$OUT/$v.json'))['ms_per_step'])") ms per step"
  python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "knn_grid" in r["Name"] or "grid_" in r["Name"]:
        print("   %-60s %8.1f us" % (r["Name"][:60], float(r["AverageNs"]) / 1e3))
P
done
