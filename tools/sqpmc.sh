#!/bin/bash
# SQ issue / wait counters per kernel of one command on the GPU box (counters in their own passes, --kernel-trace only):
#   bash tools/sqpmc.sh <tag> <python script and args...>      -> gpurun_out/sq_<tag>.txt (per-kernel sums)
set -o pipefail
TAG=$1
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/p1 -o $TAG -- python3 "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $OUT/p2 -o $TAG -- python3 "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 - <<PY > $REPO/gpurun_out/sq_$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_SALU"): calls[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    n = max(calls[(k, "SQ_WAVES")], 1)
    wc = max(d.get("SQ_WAVE_CYCLES", 1), 1)
    print("%-42s launches %4d  waves/launch %8.0f  VALU/launch %12.0f  valu_share %.3f any_share %.3f wait_any %.3f wait_inst %.3f | per launch: SALU %.3g SMEM %.3g LDS %.3g VMEM_RD %.3g VMEM_WR %.3g FLAT %.3g" % (
        k, n, d.get("SQ_WAVES", 0) / n, d.get("SQ_INSTS_VALU", 0) / n, d.get("SQ_ACTIVE_INST_VALU", 0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc,
        d.get("SQ_INSTS_SALU", 0) / n, d.get("SQ_INSTS_SMEM", 0) / n, d.get("SQ_INSTS_LDS", 0) / n, d.get("SQ_INSTS_VMEM_RD", 0) / n, d.get("SQ_INSTS_VMEM_WR", 0) / n, d.get("SQ_INSTS_FLAT", 0) / n))
PY
cat $REPO/gpurun_out/sq_$TAG.txt
