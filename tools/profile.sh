#!/bin/bash
# Profiling recipe for the GPU box (see DESIGN.md "Measurement").  Writes under gpurun_out/.
# Usage: bash tools_profile.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline"
# 1. per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
# 2. HBM traffic counters, one pass each (TCC slots: FETCH_SIZE needs 3, WRITE_SIZE 2)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1
find $OUT -name "*.csv" | head -30
