#!/bin/bash
# config-5 read shape (150 bp, band +-64) on the chr21-sized text: register budgets of the three-cell kernel, then the
# round's profile of the shape (kernel stats, FETCH/WRITE, SQ counters) under gpurun_out/prof_cfg5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for w in 3 4 5 6; do
  THM_EXT_MINW_CPL3=$w python bench.py --read-len 150 --percent 0.574 --steps 8 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/cfg5_minw$w.json 2> gpurun_out/cfg5_minw$w.err || { tail -3 gpurun_out/cfg5_minw$w.err; exit 1; }
  python3 -c "
import json; j = json.load(open('gpurun_out/cfg5_minw$w.json'))
print('minw $w: %.2f M reads/s, extend %.3f ms %s, seed %.3f' % (j['value'] / 1e6, j['roofline']['kernel_ms'], j['roofline']['kernel_ms_per_step'], j['roofline']['stage_ms']['seed']))"
done
