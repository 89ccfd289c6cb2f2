#!/bin/bash
# One gpurun call: kernel-trace stats, the counter passes (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes;
# SQ issue counters) and the plain bench / ops lines of a round.
# usage (on the GPU box, repo root): bash tools/profile_round.sh <tag>     outputs under gpurun_out/<tag>/
set -e
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 10 --warmup 3 --reuse-steps 0 > $OUT/bench_under_rocprof.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-queries 0 --reuse-steps 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-queries 0 --reuse-steps 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-queries 0 --reuse-steps 0 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write 4 $OUT/pmc_traffic.json
python tools/sq_counters.py $OUT/pmc_sq 4 $OUT/sq_counters.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
if [ "$2" != "quick" ]; then
  python tools/bench_ops.py > $OUT/ops.jsonl 2> /dev/null
  python tools/bench_knn_shapes.py > $OUT/knn_shapes.jsonl 2> /dev/null
  python tools/bench_distributions.py > $OUT/distributions.jsonl 2> /dev/null
  python tools/bench_long_lists.py > $OUT/long_lists.jsonl 2> /dev/null
  python tools/small_ops.py > $OUT/small_ops.txt 2> /dev/null
  python tools/host_overhead.py > $OUT/host_overhead.txt 2> /dev/null
  python tools/knn_small_sweep.py > $OUT/knn_small_sweep.jsonl 2> /dev/null
  python tools/fps_plan_sweep.py > $OUT/fps_plan_sweep.txt 2> /dev/null
  python tools/chamfer_probe.py 50 > $OUT/chamfer_probe.txt 2> /dev/null
fi
head -c 1500 $OUT/bench.json; echo; head -12 $OUT/kernel_stats.csv | cut -c1-100,240-330
