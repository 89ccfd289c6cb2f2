#!/bin/bash
# usage: bash tools/ab_variants.sh <variant>...   ("main" = the regular library): bench line (ms_per_step) per variant
R=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  if [ "$v" = "main" ]; then unset POINTOPS_AMD_LIB; else export POINTOPS_AMD_LIB=$R/pytorch3d_pointops_amd/lib/variants/libpointops_amd_$v.so; fi
  echo -n "$v: "; python $R/bench.py --steps 20 --warmup 5 --cpu-sample-queries 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
