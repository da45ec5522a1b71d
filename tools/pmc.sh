#!/bin/bash
# SQ instruction-mix counters for the kernels of one bench-like run (tuning aid).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o sq -- python3 $REPO/tools/perf.py ${2:-4000000} 500000 ${3:-ci} > $OUT/perf.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/sq2 -o sq2 -- python3 $REPO/tools/perf.py ${2:-4000000} 500000 ${3:-ci} > $OUT/perf2.txt 2>&1 || true
python3 - <<PY
import csv, collections, glob
for d in ("sq","sq2"):
    fs=glob.glob("$OUT/%s/*counter_collection.csv"%d)
    if not fs: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "thm::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print("%-30s %-22s n=%d mean=%.4g" % (k[0], k[1], len(v), sum(v)/len(v)))
PY
