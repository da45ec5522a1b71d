#!/bin/bash
# Per-kernel time of one command on the GPU box: bash tools/kstats.sh <tag> <python script and args...>
# Writes gpurun_out/kstats_<tag>.csv (rocprofv3 --kernel-trace --stats, the kernel_stats table) and the command's stdout.
set -o pipefail
TAG=$1
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 "$@" > $OUT/stdout.log 2> $OUT/stderr.log || { tail -20 $OUT/stderr.log; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $REPO/gpurun_out/kstats_$TAG.csv
cut -d, -f1-4 "$f" | head -14
cat $OUT/stdout.log
