#!/bin/bash
# after the depth-by-depth construction of :vegasmc starting configurations: parity, campaigns with random lane groups, timings, the bubble A/B
set -u
out=gpurun_out/r05_h
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 1500 python -m pytest tests/test_hip_spec.py tests/test_hip_parity.py tests/test_hip_steady_state.py tests/test_hip_random_configs.py -q -p no:cacheprovider > $out/pytest.txt 2>&1
tail -8 $out/pytest.txt
timeout 600 python tools/fuzz_layouts.py --lanes 0 150 > $out/fuzz_general_lanes.txt 2>&1
timeout 600 python tools/fuzz_layouts.py --carry --lanes 0 120 > $out/fuzz_carry_lanes.txt 2>&1
tail -n 1 $out/fuzz_*.txt
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1
timeout 600 python tools/spec_bench.py steps x2 vegasmc 1e4 1 16 > $out/steps_x2_vegasmc.txt 2>&1
timeout 600 python tools/spec_bench.py steps bubble vegasmc 1e6 4 16 > $out/steps_bubble_vegasmc.txt 2>&1
timeout 600 python tools/latency.py > $out/latency.txt 2>&1
for s in vegasmc mcmc; do
  timeout 900 python tools/bias_ab.py ab bubble $s 64 1e6 10 16 8 > $out/ab_bubble_${s}_b16.txt 2>&1
done
tail -n +1 $out/default.txt $out/steps_*.txt $out/latency.txt $out/ab_*.txt
