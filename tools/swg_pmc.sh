#!/bin/bash
# SQ counters of the DP micro-benchmark (tuning aid)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/swgpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o s -- python3 $REPO/tools/swg_bench.py 200000 > $OUT/out$i.txt 2>&1 || { tail -3 $OUT/out$i.txt; continue; }
done
python3 - <<PY
import csv, collections, glob
for d in ("s1","s2","s3"):
    fs=glob.glob("$OUT/%s/*counter_collection.csv"%d)
    if not fs: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "swg_batch" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][27:47], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print("%-22s %-24s n=%d vals=%s" % (k[0], k[1], len(v), " ".join("%.4g"%x for x in v[1::2])))
PY
