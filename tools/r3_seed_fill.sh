#!/bin/bash
# seed_fill_kernel's probes three ways (THM_SEED_FILL = 0 worst-case grid, 1 fixed grid striding over the listed cells,
# 2 bucketed by the leading bases of their k-mer): stage times of the default bench, then optionally of a big text
#   bash tools/r3_seed_fill.sh [genome_len for tools/big_text.py]
set -o pipefail
for m in 0 1 2; do
  THM_SEED_FILL=$m python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench THM_SEED_FILL=$m', d['value'], d['roofline']['stage_ms'])" || exit 1
done
if [ -n "$1" ]; then
  for m in 0 2 1; do
    THM_SEED_FILL=$m python3 tools/big_text.py --genome-len $1 --reads 500000 --steps 5 --oracle-reads 0 --out gpurun_out/seed_fill_big_$m.json > gpurun_out/seed_fill_big_$m.log 2>&1 || { tail -n 5 gpurun_out/seed_fill_big_$m.log; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/seed_fill_big_$m.json'))
for r in d['runs']: print('big text THM_SEED_FILL=$m', r['workload'][:30], r['reads_per_s'], r['stage_ms'])"
  done
fi
