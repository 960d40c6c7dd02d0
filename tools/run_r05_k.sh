#!/bin/bash
# after "no :vegasmc carry out of a launch on the untrained map": carried-chain parity, campaigns, cold calls on every integrand
set -u
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
