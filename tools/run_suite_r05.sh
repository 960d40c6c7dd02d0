#!/bin/bash
set -u
out=gpurun_out/r05_suite
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -60 $out/suite.txt
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold bubble 3e7 10 3 > $out/cold_bubble.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold cos 1e8 10 3 > $out/cold_cos.txt 2>&1
timeout 600 python tools/mcmc_policy.py cold c5 1e8 10 3 > $out/cold_c5.txt 2>&1
tail -n +1 $out/default.txt $out/cold_*.txt
