#!/bin/bash
# summary-kernel group sizes on the headline workload: bash tools/r3_gl.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for gl in 8 4 2 1; do
  THM_TPR=1 THM_HIT_GL=$gl bash tools/kstats.sh gl$gl $REPO/tools/perf.py 46709983 500000 ci > gpurun_out/gl$gl.txt 2>&1 || { tail -3 gpurun_out/gl$gl.txt; exit 1; }
  python3 - <<PY
import csv
for r in csv.reader(open("gpurun_out/kstats_gl$gl.csv")):
    if "hit_summary" in r[0] or "ctl_kernel" in r[0] or "dp" in r[0].lower() and "kernel" in r[0]: print("GL=$gl  %-50s calls %3s avg %8.1f us" % (r[0][:50], r[1], float(r[3]) / 1e3))
print("GL=$gl ", [l.strip() for l in open("gpurun_out/kstats_gl$gl/stdout.log") if l.startswith("ci")][0][:140])
PY
done
