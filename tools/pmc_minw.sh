#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in 5 6 8; do
  export THM_EXT_MINW=$m
  OUT=$REPO/gpurun_out/pmcm_$m; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bf.json 2>/dev/null || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bw.json 2>/dev/null || exit 1
  python3 - <<PY
import csv
def agg(p,c):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if r["Counter_Name"]==c and "extend_kernel" in r["Kernel_Name"]]
    return sum(v[1:])/max(len(v)-1,1)*1024/1e9
print("MINW $m fetch %.3f GB write %.3f GB" % (agg("$OUT/f/f_counter_collection.csv","FETCH_SIZE"), agg("$OUT/w/w_counter_collection.csv","WRITE_SIZE")))
PY
done
