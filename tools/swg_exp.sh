#!/bin/bash
# timing-only DP variants (tuning aid): which part of the column loop costs what
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in thermite_amd exp_NOSCAN exp_NONOP exp_NOTRACE; do
  export THM_LIB=$REPO/thermite_amd/_build/lib$v.so
  OUT=$REPO/gpurun_out/swgexp_$v
  mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/swg_bench.py 200000 > $OUT/out.txt 2>&1 || { tail -5 $OUT/out.txt; exit 1; }
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")) if "swg_batch" in r["Kernel_Name"]]
print("$v:", " ".join("%s=%.3f" % (r["Kernel_Name"][28:36], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6) for r in rows[1::2]))
PY
done
