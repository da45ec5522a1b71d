#!/bin/bash
# per-kernel durations of one tools_perf.py run (tuning aid): bash tools_trace.sh <tag> <ref_len> <opts>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/tools/perf.py ${2:-4000000} 500000 ${3:-ci} > $OUT/perf.txt 2>&1
cut -d, -f1-4 $OUT/t_kernel_stats.csv | head -12
tail -3 $OUT/perf.txt
