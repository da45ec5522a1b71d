#!/bin/bash
# A/B/n of several builds of the library on one GPU box: tools/abn.sh "<perf.py args>" lib1.so lib2.so ...
# (box-to-box variation is +-5 %: compare builds only within one call; two interleaved rounds)
ARGS=$1; shift
for round in 1 2; do
  for L in "$@"; do
    echo "== $L"
    THM_LIB=$PWD/$L python tools/perf.py $ARGS 2>&1 | grep -v amdgpu | grep "Mreads"
  done
done
