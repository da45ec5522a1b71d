#!/bin/bash
# Round-3 profiling recipe for the GPU box (DESIGN.md "Measurement").  Writes under gpurun_out/prof_<tag>/;
# tools/collect_r03.py turns the CSVs into the summaries kept under profiles/r03/.
#   bash tools/profile_r03.sh <tag> [stats] [hbm] [sq] [calib]      (default: all four)
# Counters are collected in their own runs (--kernel-trace only, no other trace domain), one group per pass.
set -o pipefail
TAG=${1:-r03}
shift
WHAT=${*:-stats hbm sq calib}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
# the sources the profiled library is built from (bench.py refuses counters of other sources)
python3 -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.csrc_hash())" > $OUT/csrc_hash.txt
echo "$BENCH_ARGS" > $OUT/bench_args.txt
cd /tmp && export TMPDIR=/tmp
# BENCH_ARGS: the workload (default: the headline one), e.g. BENCH_ARGS="--read-len 150 --percent 0.574" for config 5's read shape
ARGS="--steps 8 --warmup 1 --no-cpu-baseline --no-e2e $BENCH_ARGS"
for w in $WHAT; do
  case $w in
    stats)  # per-kernel time
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1 ;;
    hbm)    # HBM traffic, one pass each (TCC slots: FETCH_SIZE needs 3, WRITE_SIZE 2)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 1 ;;
    sq)     # what binds the kernels: issue and wait counters of the SQ, LDS conflicts
      rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq1 -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq1.json 2> $OUT/bench_sq1.err || exit 1
      rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -o $TAG -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/bench_sq2.err || exit 1 ;;
    calib)  # FETCH_SIZE against gathers of known size
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib -o $TAG -- python3 $REPO/tools/calib_fetch.py > $OUT/calib.json 2> $OUT/calib.err || exit 1 ;;
  esac
done
find $OUT -name "*.csv" | head -40
