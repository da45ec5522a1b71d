#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/swgtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/swg_bench.py 200000 > $OUT/out.txt 2>&1
cat $OUT/out.txt | grep xlen
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")) if "swg_batch" in r["Kernel_Name"]]
for r in rows: print(r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms", "vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"))
PY
