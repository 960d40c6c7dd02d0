#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root.  Collects, for one workload:
#   1. rocprofv3 --kernel-trace --stats  (per-kernel time)          -> gpurun_out/prof_stats_<tag>/
#   2. rocprofv3 --pmc FETCH_SIZE          (separate pass)           -> gpurun_out/prof_fetch_<tag>/
#   3. rocprofv3 --pmc WRITE_SIZE          (separate pass)           -> gpurun_out/prof_write_<tag>/
#   4. rocprofv3 --pmc SQ counters         (VALU/LDS utilisation)    -> gpurun_out/prof_sq_<tag>/
# Counter passes never combine --pmc with trace domains other than --kernel-trace (gpurun rule).
#   usage: profiles/collect.sh <tag> [bench|c4|c3|c5|bubble_mcmc|default_call] [steps]
#     bench: the default bench.py workload (C2), kernel mci_vegas_batch          -> profiles/<tag>_kernel_stats.txt, <tag>_pmc_traffic.json
#     c4   : BASELINE configs[3] on one GPU (tools/workload.py c4), mci_vegas_batch + mci_vegas_tiles
#     c3   : BASELINE configs[2] (tools/workload.py c3), mci_vegasmc_chains;   c5: BASELINE configs[4] (tools/workload.py c5), mci_mcmc_chains
set -u
TAG=${1:-r02}
WHAT=${2:-bench}
STEPS=${3:-10}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
if [ "$WHAT" = "c4" ]; then
  CMD="python tools/workload.py c4"
  KERNELS="mci_vegas_batch,mci_vegas_tiles"
elif [ "$WHAT" = "c3" ]; then        # BASELINE configs[2]: example/bubble.jl under :vegasmc, 1e8 steps per iteration
  CMD="python tools/workload.py c3"
  KERNELS="mci_vegasmc_chains"
elif [ "$WHAT" = "c5" ]; then        # BASELINE configs[4]: 4 integrals on a 12-D pool under :mcmc, automatic chain length
  CMD="python tools/workload.py c5 --niter 10 --cold"   # a cold call: ONE integrate(niter = 10) on a fresh problem, every launch in the trace
  KERNELS="mci_mcmc_chains"
elif [ "$WHAT" = "bubble_mcmc" ]; then  # the bubble diagram under :mcmc, cold: a few hundred long chains -> a group of lanes per chain (csrc/mci_spec.h)
  CMD="python tools/mcmc_policy.py cold bubble 3e7 10 1"
  KERNELS="mci_mcmc_spec,mci_mcmc_chains,k_resample_chains"
elif [ "$WHAT" = "default_call" ]; then # the reference's default call (solver = :vegasmc, neval = 1e4, 16 chains), 200 calls
  CMD="python tools/spec_bench.py default 10"
  KERNELS="mci_vegasmc_spec,mci_vegasmc_chains,mci_mcmc_spec,mci_mcmc_chains,k_finish"
else
  CMD="python bench.py --steps $STEPS --warmup 5 --passes ${PASSES:-40} --no-cpu-baseline"   # ~400 launches: the average is the steady state, not the idle ramp
  KERNELS="mci_vegas_batch"
fi
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$TAG -o stats -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/prof_stats_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch_$TAG -o fetch -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/prof_fetch_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write_$TAG -o write -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/prof_write_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/prof_sq_$TAG -o sq -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $OUT/prof_sq_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize.py $TAG "$KERNELS" "$CMD" > $OUT/profile_summary_$TAG.txt 2>&1; cp profiles/${TAG}_kernel_stats.txt profiles/${TAG}_pmc_traffic.json $OUT/ 2>/dev/null
# (the raw rocpd databases are tens of MiB per pass and gpurun brings back at most 64 MiB: only the summaries travel)
rm -rf $OUT/prof_stats_$TAG $OUT/prof_fetch_$TAG $OUT/prof_write_$TAG $OUT/prof_sq_$TAG
tail -60 $OUT/profile_summary_$TAG.txt
