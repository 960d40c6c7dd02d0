#!/bin/bash
# GPU-box batches behind the numbers in profiles/ (one parameterised script; run as `gpurun -- bash tools/run_batch.sh <batch> [args]`).
# Every batch writes under gpurun_out/<batch>/; what is to be judged is copied into profiles/ by hand afterwards.
#   round 6:  ab_exec_mask | midsize | r06_collect | r06_recollect | r06_fuzz
#   round 5 (kept as they ran, cited by profiles/README.md): r05_<name>
set -u
batch=${1:-help}; shift || true
out=gpurun_out/$batch; mkdir -p $out
python -c "import torch" >/dev/null 2>&1   # (pages the image in: the first import of a fresh box takes a minute)
case "$batch" in
ab_exec_mask)
# si-optimize-exec-masking-pre-ra off for EVERY JIT unit (MCI_JIT_FLAGS names the switch, mci_jit.h) against the
# default pipeline: the headline, C2 on 16 grids, C4, C3 :vegasmc, C5 :mcmc / :vegasmc / :vegas -- each variant on its own cold kernel cache,
# two repetitions, interleaved
for rep in 1 2; do
for v in default off; do   # default = LLVM's default pipeline (the pass on): forced, since mci_jit.h now switches it off for every unit
  flags="-mllvm -amdgpu-opt-exec-mask-pre-ra=1"; [ $v = off ] && flags="-mllvm -amdgpu-opt-exec-mask-pre-ra=0"
  export MCI_KERNEL_CACHE=/tmp/kc_$v MCI_JIT_FLAGS="$flags"
  timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
  timeout 900 python tools/bench_configs.py c2i c4 c3mc c5 > $out/configs_${v}_$rep.txt 2>&1
  timeout 300 python tools/spec_bench.py default > $out/default_${v}_$rep.txt 2>&1
done; done
unset MCI_KERNEL_CACHE MCI_JIT_FLAGS
python - <<'PY'
import json, glob, re
for v in ("default", "off"):
    for f in sorted(glob.glob("gpurun_out/ab_exec_mask/bench_%s_*.json" % v)):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1]); print(v, "headline", j["value"], "Msamples/s, kernel", j["roofline"]["kernel_ms_avg"], "ms")
        except Exception as e:
            print(v, f, "unreadable", e)
    for f in sorted(glob.glob("gpurun_out/ab_exec_mask/configs_%s_*.txt" % v)):
        for ln in open(f):
            m = re.search(r"TRAINED:\s+([0-9.]+) Msamples/s\s+kernel ([0-9.]+) ms", ln)
            if m: print(v, name, "trained", m.group(1), "Msamples/s kernel", m.group(2), "ms")
            elif "COLD" in ln: name = ln[:28].strip()
    for f in sorted(glob.glob("gpurun_out/ab_exec_mask/default_%s_*.txt" % v)):
        print(v, "default call:", open(f).read().strip().splitlines()[-1][:200])
PY
;;
r06_collect)
# the final batch of round 6 on the final code: GPU suite (timed), smoke, bench line + rocprofv3 / PMC of the same command, every other
# configuration profiled, cold and trained rates, latencies, the validation matrix
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rs --durations=12 > $out/suite.txt 2>&1; tail -3 $out/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python bench.py > $out/r06_bench_line.json 2> $out/bench.err
bash profiles/collect.sh r06 bench > $out/collect_bench.log 2>&1
for w in c3 c4 c5 bubble_mcmc default_call; do bash profiles/collect.sh r06_$w $w > $out/collect_$w.log 2>&1; done
cp profiles/r06*_kernel_stats.txt profiles/r06*_pmc_traffic.json $out/ 2>/dev/null
sleep 20   # (the profiler passes above have just freed tens of GB: the driver wipes VRAM on release, and a 4.8 GB hipMalloc that lands on pages
           # still being wiped waits for them -- profiles/r06_other_configs.txt; a cold call is quoted on an idle device)
timeout 900 python tools/bench_configs.py > $out/other_configs.txt 2>&1
timeout 600 python tools/latency.py > $out/latency.txt 2>&1
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1
timeout 900 python tools/validation_matrix.py > $out/validation_matrix.txt 2>&1
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_collect/r06_bench_line.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(j["value"], j["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("frac_flat_2cycle"), r["frac_self_calibrated"], r["clock"]["sclk_mhz_avg"], r["traffic"], j["config"].get("code_object"))
PY
;;
r06_recollect)
# after the last change of the generated source in round 6 (a Continuous leaf's bounds left out of it: every code object got a new name, the
# machine code is the same): GPU suite, smoke, bench line and the rocprofv3 / PMC summaries of every profiled workload again, so that
# profiles/r06_* name the code objects the committed tree produces
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/suite.txt 2>&1; tail -2 $out/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
bash profiles/collect.sh r06 bench > $out/collect_bench.log 2>&1
timeout 900 python bench.py > $out/r06_bench_line.json 2> $out/bench.err
for w in c3 c4 c5 bubble_mcmc default_call; do bash profiles/collect.sh r06_$w $w > $out/collect_$w.log 2>&1; done
cp profiles/r06*_kernel_stats.txt profiles/r06*_pmc_traffic.json $out/ 2>/dev/null
python -c "import json; d = json.load(open('$out/r06_bench_line.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), d['config']['code_object'])"
;;
r06_fuzz)
# randomised parity campaigns on the final code of round 6, cold kernel cache: every new several-lanes-per-chain code object goes through its
# self-check (mci_host_jit.h spec_self_check) -- a campaign is also a count of its false alarms (stderr: "does not reproduce")
export MCI_KERNEL_CACHE=/tmp/kc_fuzz
timeout 1500 python tools/fuzz_layouts.py --lanes 1000 120 > $out/fuzz_general_lanes.txt 2> $out/fuzz_general_lanes.err; tail -2 $out/fuzz_general_lanes.txt
timeout 1200 python tools/fuzz_layouts.py --carry --lanes 1000 100 > $out/fuzz_carry_lanes.txt 2> $out/fuzz_carry_lanes.err; tail -2 $out/fuzz_carry_lanes.txt
timeout 900 python tools/fuzz_layouts.py 1000 60 > $out/fuzz_general.txt 2> $out/fuzz_general.err; tail -2 $out/fuzz_general.txt
timeout 600 python tools/fuzz_layouts.py --pipe 1000 30 > $out/fuzz_pipe.txt 2> $out/fuzz_pipe.err; tail -2 $out/fuzz_pipe.txt
echo "self-check alarms:"; grep -c "does not reproduce" $out/*.err
echo "self-check markers written:"; ls /tmp/kc_fuzz/*.ok 2>/dev/null | wc -l; echo "code objects:"; ls /tmp/kc_fuzz/*.hsaco | wc -l
unset MCI_KERNEL_CACHE
;;
midsize)
# the fixed cost of a mid-size launch of the headline layout -- the copy summation with conflict-free reads + row_shr adds
# (-DMCI_COPY_SUM_DPP=1), 16-byte zeroing stores (-DMCI_ZERO_B128=1), fewer histogram copies -- each variant twice, interleaved
for rep in 1 2; do
python tools/midsize_c2.py "round 5 epilogue"
MCI_JIT_FLAGS="-DMCI_COPY_SUM_DPP=1" python tools/midsize_c2.py "dpp sum"
MCI_JIT_FLAGS="-DMCI_ZERO_B128=1" python tools/midsize_c2.py "b128 zero"
MCI_JIT_FLAGS="-DMCI_COPY_SUM_DPP=1 -DMCI_ZERO_B128=1" python tools/midsize_c2.py "dpp sum + b128 zero"
MCI_JIT_FLAGS="-DMCI_COPY_SUM_DPP=1 -DMCI_ZERO_B128=1" python tools/midsize_c2.py --copies 4 "both, 4 copies"
MCI_JIT_FLAGS="-DMCI_COPY_SUM_DPP=1 -DMCI_ZERO_B128=1" python tools/midsize_c2.py --copies 2 "both, 2 copies"
MCI_JIT_FLAGS="-DMCI_COPY_SUM_DPP=1 -DMCI_ZERO_B128=1" python tools/midsize_c2.py --copies 1 "both, 1 copy"
python tools/midsize_c2.py --copies 2 "round 5, 2 copies"
done 2>&1 | tee $out/midsize.txt
;;
r05_bias)
# GPU box: the three bias flags of VERDICT r04 at their own configurations (tools/bias_ab.py) -> gpurun_out/r05_bias/
out=gpurun_out/r05_bias
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
NS=${NS:-256}
# (a) cold integrate(neval = 1e8, niter = 10) of the chain solvers' BASELINE configurations, iteration by iteration
for cs in "bubble vegasmc" "c5 vegasmc" "c5 mcmc"; do
  set -- $cs
  timeout 900 python tools/bias_ab.py full $1 $2 ${NS_FULL:-128} 1e8 10 16 4 > $out/full_$1_$2_1e8.txt 2>&1
done
# (b) the reference's chain against the automatic chains at the same (neval = 1e6, block)
for cs in "c5 vegasmc" "c5 mcmc" "sphere2 vegasmc" "sphere2 mcmc" "bubble vegasmc" "bubble mcmc"; do
  set -- $cs
  for blk in 16 64; do
    timeout 1500 python tools/bias_ab.py ab $1 $2 $NS 1e6 10 $blk 8 > $out/ab_$1_$2_b$blk.txt 2>&1
  done
done
tail -n +1 $out/*.txt
;;
r05_a4)
out=gpurun_out/r05_a4
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
for cs in x2 log sphere2 hyper cos c5 bubble; do
  timeout 300 python tools/bias_ab.py full $cs vegasmc 64 1e7 10 16 4 > $out/full_${cs}_vegasmc_1e7.txt 2>&1
done
for f in $out/full_*; do grep -v "resource_tracker\|warnings.warn" $f | head -4; grep "iteration  2 \|iteration  3 " $f; done
;;
r05_collect)
# final r05 collection: bench line + rocprofv3 / PMC summaries of the same command; other configurations; the GPU suite on the final code
out=gpurun_out/r05_f
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python bench.py > $out/r05_bench_line.json 2> $out/bench.err
tail -c 600 $out/r05_bench_line.json; echo
bash profiles/collect.sh r05 bench > $out/collect_bench.log 2>&1
bash profiles/collect.sh r05_c3 c3 > $out/collect_c3.log 2>&1
bash profiles/collect.sh r05_c4 c4 > $out/collect_c4.log 2>&1
bash profiles/collect.sh r05_c5 c5 > $out/collect_c5.log 2>&1
bash profiles/collect.sh r05_bubble_mcmc bubble_mcmc > $out/collect_bubble.log 2>&1
bash profiles/collect.sh r05_default_call default_call > $out/collect_default.log 2>&1
cp profiles/r05*_kernel_stats.txt profiles/r05*_pmc_traffic.json $out/ 2>/dev/null
timeout 600 python tools/midsize_sweep.py gauss6 > $out/midsize_gauss6.txt 2>&1
timeout 900 python tools/bench_configs.py > $out/other_configs.txt 2>&1
du -sh gpurun_out
ls $out
;;
r05_d)
out=gpurun_out/r05_d
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python tools/midsize_sweep.py > $out/midsize.txt 2>&1
# opt-in contraction, measured first: the whole translation unit under -ffp-contract=fast (the later flag wins)
timeout 400 python bench.py --min-seconds 2 --no-cpu-baseline > $out/bench_contract_off.json 2> $out/bench_contract_off.err
MCI_JIT_FLAGS="-ffp-contract=fast" timeout 400 python bench.py --min-seconds 2 --no-cpu-baseline > $out/bench_contract_fast.json 2> $out/bench_contract_fast.err
timeout 300 python tools/workload.py c4 --niter 8 > $out/c4_contract_off.txt 2>&1
MCI_JIT_FLAGS="-ffp-contract=fast" timeout 300 python tools/workload.py c4 --niter 8 > $out/c4_contract_fast.txt 2>&1
MCI_JIT_FLAGS="-DMCI_GATHER_X4=1" timeout 300 python tools/workload.py c4 --niter 8 > $out/c4_gather_x4.txt 2>&1
timeout 300 python tools/workload.py c4 --niter 8 > $out/c4_contract_off_b.txt 2>&1
timeout 600 python tools/latency.py > $out/latency.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_d/bench_contract_*.json")):
    j=json.load(open(f)); r=j["roofline"]; print(f, j["value"], j["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["clock"]["sclk_mhz_avg"], r["valu_datasheet"]["cycles_per_wave_sample"], j["estimate"]["deviation_sigma"])
PY
tail -n +1 $out/midsize.txt $out/c4_*.txt $out/latency.txt
;;
r05_e)
out=gpurun_out/r05_e
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -40 $out/suite.txt
timeout 600 python tools/midsize_sweep.py x2y2 gauss6 > $out/midsize_after.txt 2>&1
timeout 600 python tools/latency.py > $out/latency.txt 2>&1
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1
for cs in "x2 vegasmc 1e4 1" "x2 mcmc 1e4 1" "bubble mcmc 3e6 16" "cos mcmc 1e7 32" "c5 mcmc 1e7 64" "bubble vegasmc 1e6 4"; do
  set -- $cs
  timeout 600 python tools/spec_bench.py steps $1 $2 $3 $4 16 > $out/steps_$1_$2.txt 2>&1
done
timeout 600 python tools/mcmc_policy.py cold bubble 3e7 10 3 > $out/cold_bubble.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold cos 1e8 10 3 > $out/cold_cos.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold c5 1e8 10 3 > $out/cold_c5.txt 2>&1
timeout 900 python tools/bench_configs.py > $out/other_configs.txt 2>&1
tail -n +1 $out/midsize_after.txt $out/latency.txt $out/default.txt $out/steps_*.txt $out/cold_*.txt $out/other_configs.txt
;;
r05_final)
# the last GPU batch of round 5, on the final code objects: GPU suite, smoke, bench line + rocprofv3 / PMC of the same command, C3 / C5 :vegasmc
# and the default call re-profiled (their kernels changed late), every configuration cold and trained
out=gpurun_out/r05_final
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -4 $out/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python bench.py > $out/r05_bench_line.json 2> $out/bench.err
bash profiles/collect.sh r05 bench > $out/collect_bench.log 2>&1
bash profiles/collect.sh r05_c3 c3 > $out/collect_c3.log 2>&1
bash profiles/collect.sh r05_default_call default_call > $out/collect_default.log 2>&1
cp profiles/r05_kernel_stats.txt profiles/r05_pmc_traffic.json profiles/r05_c3_kernel_stats.txt profiles/r05_c3_pmc_traffic.json profiles/r05_default_call_kernel_stats.txt profiles/r05_default_call_pmc_traffic.json $out/
timeout 900 python tools/bench_configs.py > $out/other_configs.txt 2>&1
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05_final/r05_bench_line.json")); r=j["roofline"]
print(j["value"], j["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["frac_self_calibrated"], r["clock"]["sclk_mhz_avg"], r["traffic"], j["config"]["code_object"])
print(json.load(open("gpurun_out/r05_final/r05_pmc_traffic.json"))["code_object"])
PY
cat $out/default.txt
;;
r05_floors)
# automatic :vegasmc chains started afresh (iterations 1 and 2 of a cold call): their length, in burn-in floors, and the part of them that is not
# measured against the bias of the second iteration and the time per run (profiles/r05_bias.txt A5; csrc/mci_debug.h fresh_floors, fresh_burnin_pct)
out=gpurun_out/r05_floors
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
for fl in 8 16 32 64; do
  BIAS_FRESH_FLOORS=$fl timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_fl$fl.txt 2>&1
done
for fl in 8 32 64; do
  BIAS_FRESH_FLOORS=$fl timeout 400 python tools/bias_ab.py full bubble vegasmc 16 1e8 10 16 2 > $out/bubble_fl$fl.txt 2>&1
  BIAS_FRESH_FLOORS=$fl timeout 400 python tools/bias_ab.py full c5 vegasmc 16 1e8 10 16 2 > $out/c5_fl$fl.txt 2>&1
  BIAS_FRESH_FLOORS=$fl timeout 300 python tools/bias_ab.py full cos vegasmc 32 1e7 10 16 4 > $out/cos_fl$fl.txt 2>&1
done
for b in 25 50 75; do
  BIAS_FRESH_BURNIN=$b timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_burn$b.txt 2>&1
done
BIAS_FRESH_FLOORS=16 BIAS_FRESH_BURNIN=50 timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_fl16_burn50.txt 2>&1
BIAS_FRESH_FLOORS=32 BIAS_FRESH_BURNIN=50 timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_fl32_burn50.txt 2>&1
for f in $out/*.txt; do echo "== $f"; grep -v "resource_tracker\|warnings.warn" $f | head -9; done
;;
r05_g)
# r05 evidence on the final code: randomised parity campaigns (with random lane groups / trees), the validation matrix, the GPU suite
out=gpurun_out/r05_g
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python tools/fuzz_layouts.py 0 150 > $out/fuzz_general.txt 2>&1
timeout 900 python tools/fuzz_layouts.py --lanes 0 150 > $out/fuzz_general_lanes.txt 2>&1
timeout 900 python tools/fuzz_layouts.py --carry 0 120 > $out/fuzz_carry.txt 2>&1
timeout 900 python tools/fuzz_layouts.py --carry --lanes 0 120 > $out/fuzz_carry_lanes.txt 2>&1
timeout 300 python tools/fuzz_layouts.py --persist 0 100 > $out/fuzz_persist.txt 2>&1
timeout 300 python tools/fuzz_layouts.py --pipe 0 40 > $out/fuzz_pipe.txt 2>&1
tail -n 1 $out/fuzz_*.txt
timeout 2400 python tools/validation_matrix.py 16 1e7 > $out/validation_matrix.txt 2>&1
cat $out/validation_matrix.txt
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -5 $out/suite.txt
;;
r05_i)
out=gpurun_out/r05_i
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 600 python -m pytest tests/test_hip_spec.py -q -p no:cacheprovider > $out/pytest_spec.txt 2>&1
tail -4 $out/pytest_spec.txt
timeout 300 python tools/fuzz_layouts.py --lanes 0 60 > $out/fuzz_general_lanes.txt 2>&1
timeout 300 python tools/fuzz_layouts.py --carry --lanes 0 50 > $out/fuzz_carry_lanes.txt 2>&1
tail -n 1 $out/fuzz_*.txt
timeout 200 python tools/spec_bench.py default > $out/default.txt 2>&1
timeout 300 python tools/spec_bench.py steps x2 vegasmc 1e4 1 16 > $out/steps_x2_vegasmc.txt 2>&1
timeout 300 python tools/spec_bench.py steps bubble vegasmc 1e6 4 16 > $out/steps_bubble_vegasmc.txt 2>&1
tail -n +1 $out/default.txt $out/steps_*.txt
;;
r05_j)
# final code: GPU suite, bench line, latency tables, the default-call profile again (its kernel changed), the bubble A/B of profiles/r05_bias.txt B
out=gpurun_out/r05_j
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -5 $out/suite.txt
timeout 600 python bench.py > $out/bench_line.json 2> $out/bench.err
MCI_BENCH_FORCE_COMM=1 timeout 600 python bench.py --min-seconds 2 --no-cpu-baseline > $out/bench_line_forced_comm.json 2> $out/bench_fc.err
timeout 600 python tools/latency.py > $out/latency.txt 2>&1
bash profiles/collect.sh r05_default_call default_call > $out/collect_default.log 2>&1
cp profiles/r05_default_call_kernel_stats.txt profiles/r05_default_call_pmc_traffic.json $out/
timeout 400 python tools/bias_ab.py ab bubble vegasmc 32 1e6 10 16 1 > $out/ab_bubble_vegasmc_b16.txt 2>&1
timeout 600 python tools/bias_ab.py ab bubble mcmc 32 1e6 10 16 1 > $out/ab_bubble_mcmc_b16.txt 2>&1
tail -n +1 $out/latency.txt $out/ab_*.txt
python - <<'PY'
import json
for f in ("gpurun_out/r05_j/bench_line.json","gpurun_out/r05_j/bench_line_forced_comm.json"):
    j=json.load(open(f)); r=j["roofline"]; print(f, j["value"], j["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["clock"]["sclk_mhz_avg"], j["comm"]["kind"], j["comm"].get("collectives"))
PY
;;
r05_k)
# after "no :vegasmc carry out of a launch on the untrained map": carried-chain parity, campaigns, cold calls on every integrand
out=gpurun_out/r05_k
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_steady_state.py tests/test_hip_spec.py tests/test_hip_random_configs.py tests/test_distributed_gloo.py -m gpu -q -p no:cacheprovider -k "carried or carry or two_ranks or random" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
timeout 600 python tools/fuzz_layouts.py --carry 0 80 > $out/fuzz_carry.txt 2>&1
timeout 600 python tools/fuzz_layouts.py --carry --lanes 0 80 > $out/fuzz_carry_lanes.txt 2>&1
tail -n 1 $out/fuzz_*.txt
for cs in log x2 sphere2 hyper cos c5 bubble; do
  timeout 300 python tools/bias_ab.py full $cs vegasmc 64 1e7 10 16 4 > $out/full_${cs}_vegasmc_1e7.txt 2>&1
done
for cs in c5 bubble; do
  timeout 300 python tools/bias_ab.py full $cs vegasmc 64 1e8 10 16 4 > $out/full_${cs}_vegasmc_1e8.txt 2>&1
done
for f in $out/full_*; do grep -v "resource_tracker\|warnings.warn" $f | head -3; grep "iteration  2 \|iteration  3 " $f; done
timeout 600 python tools/bench_configs.py 2>&1 | grep -A1 "vegasmc" > $out/other_vegasmc.txt; cat $out/other_vegasmc.txt | cut -c1-260
;;
r05_log)
out=gpurun_out/r05_log
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
BIAS_CARRY=off timeout 300 python tools/bias_ab.py full log vegasmc 64 1e7 10 16 4 > $out/log_carry_off.txt 2>&1
BIAS_NCHAIN=64 timeout 300 python tools/bias_ab.py full log vegasmc 64 1e7 10 16 4 > $out/log_nchain64.txt 2>&1
BIAS_NCHAIN=1 timeout 600 python tools/bias_ab.py full log vegasmc 32 1e7 10 16 4 > $out/log_nchain1.txt 2>&1
timeout 300 python tools/bias_ab.py full log vegas 64 1e7 10 16 4 > $out/log_vegas.txt 2>&1
timeout 300 python tools/bias_ab.py full log mcmc 64 1e7 10 16 4 > $out/log_mcmc.txt 2>&1
for f in $out/*.txt; do grep -v "resource_tracker\|warnings.warn" $f | head -16; done
;;
r05_o3fix)
# the several-lanes-per-chain units at -O3 without si-optimize-exec-masking-pre-ra (csrc/mci_jit.h)
out=gpurun_out/r05_o3fix
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 200 python tools/repro_case.py 205 2>&1 | grep "^lanes" | cut -c1-200
timeout 900 python -m pytest tests/test_hip_spec.py tests/test_hip_steady_state.py -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
timeout 300 python tools/fuzz_layouts.py --carry --lanes 200 30 > $out/fuzz_a.txt 2>&1; tail -n 1 $out/fuzz_a.txt
timeout 300 python tools/spec_bench.py default 2>&1 | grep -v "warn\|resource"
timeout 200 python tools/mcmc_policy.py cold cos 1e8 10 2 2>&1 | grep -v "warn\|resource" | tail -1
timeout 200 python tools/mcmc_policy.py cold bubble 3e7 10 3 2>&1 | grep -v "warn\|resource" | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
;;
r05_onepass)
# :mcmc groups with ONE proposal pass per trip (csrc/mci_spec.h): parity first, then the regimes profiles/r05_spec.txt holds
out=gpurun_out/r05_onepass
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_spec.py -m gpu -q -p no:cacheprovider -x > $out/pytest_spec.txt 2>&1
tail -4 $out/pytest_spec.txt
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1; grep -v "warn\|resource" $out/default.txt
for rep in 1 2; do
timeout 300 python tools/mcmc_policy.py cold bubble 3e7 10 1 2>&1 | grep -v "warn\|resource" | tail -2
timeout 300 python tools/mcmc_policy.py cold cos 1e8 10 1 2>&1 | grep -v "warn\|resource" | tail -2
done
timeout 300 python tools/mcmc_policy.py cold c5 1e8 10 1 2>&1 | grep -v "warn\|resource" | tail -2
timeout 300 python tools/spec_bench.py steps bubble mcmc 1e6 1 16 > $out/steps_bubble.txt 2>&1; grep -v "warn\|resource" $out/steps_bubble.txt
timeout 300 python tools/spec_bench.py steps cos mcmc 1e6 1 16 > $out/steps_cos.txt 2>&1; grep -v "warn\|resource" $out/steps_cos.txt
;;
r05_traces)
out=gpurun_out/r05_c
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_battery.py -q -p no:cacheprovider -k "closure or bias or unbiased or cold_vegasmc" > $out/pytest_a.txt 2>&1
tail -30 $out/pytest_a.txt
timeout 300 python tools/mcmc_policy.py trace c5 1e8 10 > $out/trace_c5_auto.txt 2>&1
POLICY_LANES=1 timeout 300 python tools/mcmc_policy.py trace c5 1e8 10 > $out/trace_c5_lanes1.txt 2>&1
timeout 300 python tools/mcmc_policy.py trace bubble 3e7 10 > $out/trace_bubble_auto.txt 2>&1
timeout 300 python tools/mcmc_policy.py trace cos 1e8 10 > $out/trace_cos_auto.txt 2>&1
for cs in c5 bubble; do
  timeout 600 python tools/bias_ab.py full $cs vegasmc 64 1e8 10 16 4 > $out/full_${cs}_vegasmc_resampled.txt 2>&1
done
timeout 600 python tools/bias_ab.py full c5 vegasmc 64 1e6 10 16 4 > $out/full_c5_vegasmc_1e6_resampled.txt 2>&1
timeout 900 python bench.py --min-seconds 3 > $out/bench.json 2> $out/bench.err
tail -n +1 $out/trace_*.txt $out/full_*.txt; python -c "
import json;j=json.load(open('$out/bench.json'));r=j['roofline'];print(j['value'],j['ms_per_step'],r['frac'],r.get('frac_self_calibrated'),r['clock'],r['valu_datasheet'].get('issue_ns_per_wave_sample'),r['kernel_ms_avg'])"
;;
*)
grep -E "^[a-z0-9_]+\)$" "$0" | tr -d ")" | tr "\n" " "; echo
;;
esac
