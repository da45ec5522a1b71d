#!/bin/bash
# the round's profiles in one GPU-box call: headline workload (tag r03) and config 5's read shape (tag cfg5)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/profile_r03.sh r03 stats hbm sq > gpurun_out/profile_r03.log 2>&1 || { tail -5 gpurun_out/profile_r03.log; exit 1; }
echo "r03 done"
BENCH_ARGS="--read-len 150 --percent 0.574" bash tools/profile_r03.sh cfg5 stats hbm sq > gpurun_out/profile_cfg5.log 2>&1 || { tail -5 gpurun_out/profile_cfg5.log; exit 1; }
echo "cfg5 done"
