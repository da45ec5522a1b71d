#!/bin/bash
# One GPU-box call: per-kernel times of the headline workload on the problem-parallel path and on the wave-per-read
# kernels alone (tools/perf.py under rocprofv3 --kernel-trace --stats), then whatever tests are named.
#   bash tools/r3_measure.sh <tag> [pytest args...]
set -o pipefail
TAG=${1:-m}
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/kstats.sh ${TAG}_tpr $REPO/tools/perf.py 46709983 500000 ci > gpurun_out/${TAG}_tpr.txt 2>&1 || { tail -5 gpurun_out/${TAG}_tpr.txt; exit 1; }
THM_TPR=0 bash tools/kstats.sh ${TAG}_wave $REPO/tools/perf.py 46709983 500000 ci > gpurun_out/${TAG}_wave.txt 2>&1 || { tail -5 gpurun_out/${TAG}_wave.txt; exit 1; }
python3 - <<PY
import csv
for t in ("${TAG}_tpr", "${TAG}_wave"):
    print("==", t)
    for r in list(csv.reader(open("gpurun_out/kstats_%s.csv" % t)))[1:14]:
        print("  %-70s calls %4s avg %9.1f us" % (r[0][:70], r[1], float(r[3]) / 1e3))
    print(open("gpurun_out/%s.txt" % t).read().split("\n", 14)[-1] if False else "".join(l for l in open("gpurun_out/kstats_%s/stdout.log" % t)))
PY
if [ $# -gt 0 ]; then
  timeout -k 10 1000 python -m pytest "$@" -x -q > gpurun_out/${TAG}_tests.log 2>&1
  tail -6 gpurun_out/${TAG}_tests.log
fi
