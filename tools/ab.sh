#!/bin/bash
# A/B of two builds of the library on the same GPU box: tools/ab.sh <libA.so> <libB.so> [perf.py args]
# (box-to-box variation is +-5 %: compare builds only within one call; each build runs twice, interleaved)
A=$1; B=$2; shift; shift
for L in $A $B $A $B; do
  echo "== $L"
  THM_LIB=$PWD/$L python tools/perf.py ${*:-46709983 500000 ci} 2>&1 | grep -v amdgpu | grep "Mreads"
done
