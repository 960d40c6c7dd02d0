#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root.  Collects, for the default bench workload:
#   1. rocprofv3 --kernel-trace --stats  (per-kernel time)          -> gpurun_out/prof_stats/
#   2. rocprofv3 --pmc FETCH_SIZE          (separate pass)           -> gpurun_out/prof_fetch/
#   3. rocprofv3 --pmc WRITE_SIZE          (separate pass)           -> gpurun_out/prof_write/
#   4. rocprofv3 --pmc SQ counters         (VALU/LDS utilisation)    -> gpurun_out/prof_sq/
# Counter passes never combine --pmc with trace domains other than --kernel-trace (gpurun rule).
set -u
TAG=${1:-r01}
STEPS=${2:-10}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
BENCH="python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$TAG -o stats -- bash -c "cd $GRAFT_REPO_ROOT && $BENCH" > $OUT/prof_stats_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch_$TAG -o fetch -- bash -c "cd $GRAFT_REPO_ROOT && $BENCH" > $OUT/prof_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write_$TAG -o write -- bash -c "cd $GRAFT_REPO_ROOT && $BENCH" > $OUT/prof_write_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/prof_sq_$TAG -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $BENCH" > $OUT/prof_sq_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize.py $TAG > $OUT/profile_summary_$TAG.txt 2>&1; cp profiles/${TAG}_kernel_stats.txt profiles/${TAG}_pmc_traffic.json $OUT/ 2>/dev/null
tail -40 $OUT/profile_summary_$TAG.txt
