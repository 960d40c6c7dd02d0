#!/bin/bash
set -u
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
