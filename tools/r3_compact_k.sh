#!/bin/bash
# compact stage with 1, 2 and 4 reads in flight per 16-lane group (THM_COMPACT_K): stage times of the default bench
set -o pipefail
for k in 1 2 4; do
  THM_COMPACT_K=$k python3 bench.py --steps 12 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$k', d['value'], d['roofline']['stage_ms'])" || exit 1
done
