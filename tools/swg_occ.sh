#!/bin/bash
# DP micro-benchmark at several occupancies (tuning aid): THM_SWG_BPC = resident workgroups per CU
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for bpc in 1 2 4 8; do
  export THM_SWG_BPC=$bpc
  OUT=$REPO/gpurun_out/swgocc_$bpc
  mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/swg_bench.py 200000 > $OUT/out.txt 2>&1 || exit 1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")) if "swg_batch" in r["Kernel_Name"]]
print("bpc $bpc:", " ".join("%s=%.3f" % (r["Kernel_Name"][28:46], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6) for r in rows[1::2]))
PY
done
