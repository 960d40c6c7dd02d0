#!/bin/bash
set -u
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
