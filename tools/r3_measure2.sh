#!/bin/bash
# bash tools/r3_measure2.sh <tag> [pytest args...]: per-launch timeline + SQ counters of the headline workload on the problem-parallel path
set -o pipefail
TAG=${1:-m}
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
THM_TPR=1 bash tools/kstats.sh ${TAG}_tpr $REPO/tools/perf.py 46709983 500000 ci > gpurun_out/${TAG}_tpr.txt 2>&1 || { tail -5 gpurun_out/${TAG}_tpr.txt; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kstats_${TAG}_tpr/${TAG}_tpr_kernel_trace.csv")))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]) for r in rows)
i0 = [i for i, k in enumerate(ks) if 'sanitize' in k[2]][-1]
t0 = ks[i0][0]
for k in ks[i0:]:
    if 'rocclr' in k[2] or 'scan_' in k[2] or 'widen' in k[2]: continue
    print("%8.1f %8.1f us  %s" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[2]))
print(open("gpurun_out/kstats_${TAG}_tpr/stdout.log").read())
PY
THM_TPR=1 bash tools/sqpmc.sh ${TAG} $REPO/tools/perf.py 46709983 500000 ci > /dev/null 2>&1; grep -E "ctl|dp_kernel|extend_kernel" gpurun_out/sq_${TAG}.txt
if [ $# -gt 0 ]; then
  timeout -k 10 1000 python -m pytest "$@" -x -q > gpurun_out/${TAG}_tests.log 2>&1
  tail -6 gpurun_out/${TAG}_tests.log
fi
