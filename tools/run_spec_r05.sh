#!/bin/bash
# GPU box: first contact of the several-lanes-per-chain kernels -> gpurun_out/r05_spec/
set -u
out=gpurun_out/r05_spec
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 1500 python -m pytest tests/test_hip_spec.py -x -q -p no:cacheprovider > $out/pytest_spec.txt 2>&1
tail -15 $out/pytest_spec.txt
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1
timeout 600 python tools/spec_bench.py steps x2 vegasmc 1e4 1 16 > $out/steps_x2_vegasmc_1e4.txt 2>&1
timeout 600 python tools/spec_bench.py steps x2 mcmc 1e4 1 16 > $out/steps_x2_mcmc_1e4.txt 2>&1
timeout 900 python tools/spec_bench.py steps bubble mcmc 3e6 16 16 > $out/steps_bubble_mcmc.txt 2>&1
timeout 900 python tools/spec_bench.py steps cos mcmc 1e7 32 16 > $out/steps_cos_mcmc.txt 2>&1
timeout 900 python tools/spec_bench.py steps c5 mcmc 1e7 64 16 > $out/steps_c5_mcmc.txt 2>&1
timeout 900 python tools/spec_bench.py steps bubble vegasmc 1e6 4 16 > $out/steps_bubble_vegasmc.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold bubble 3e7 10 2 > $out/cold_bubble.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold cos 1e8 10 2 > $out/cold_cos.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold c5 1e8 10 2 > $out/cold_c5.txt 2>&1
tail -n +1 $out/default.txt $out/steps_*.txt $out/cold_*.txt
