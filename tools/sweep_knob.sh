#!/bin/bash
# usage: bash tools/sweep_knob.sh <knob> v1 v2 ...   (bench ms_per_step per POINTOPS_DEBUG=<knob>=<v>[,$EXTRA])
R=${GRAFT_REPO_ROOT:-$PWD}
K=$1; shift
for v in "$@"; do
  export POINTOPS_DEBUG="$K=$v${EXTRA:+,$EXTRA}"
  echo -n "$POINTOPS_DEBUG: "; python $R/bench.py --steps 20 --warmup 5 --cpu-sample-queries 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
