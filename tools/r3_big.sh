#!/bin/bash
# Big-text stage costs with counters (round-3 verdict item 4): kernel stats, then FETCH_SIZE / WRITE_SIZE and the
# vector-L1 translation counters in passes of their own, all on one text size.   bash tools/r3_big.sh <genome_len>
set -o pipefail
G=${1:-300000000}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/big_$G
mkdir -p $OUT
ARGS="$REPO/tools/big_text.py --genome-len $G --reads 500000 --steps 5 --oracle-reads 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o big -- python3 $ARGS --out $OUT/run_stats.json > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
cp "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o big -- python3 $ARGS --out $OUT/run_fetch.json > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o big -- python3 $ARGS --out $OUT/run_write.json > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
echo "write pass done"
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $OUT/tlb -o big -- python3 $ARGS --out $OUT/run_tlb.json > $OUT/tlb.log 2>&1 || { tail -5 $OUT/tlb.log; echo "tlb pass failed (counters not available?)"; }
python3 - <<PY
import csv, glob, collections
for tag in ("fetch", "write", "tlb"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        print(tag, k, {c: (len(v), round(max(v), 1), round(sorted(v)[len(v) // 2], 1)) for c, v in d.items()})
PY
grep -h "stage_ms\|reads_per_s" $OUT/run_stats.json | cut -c1-1500
rm -rf $OUT/stats $OUT/fetch/*/*agent* 2>/dev/null
