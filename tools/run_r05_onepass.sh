# :mcmc groups with ONE proposal pass per trip (csrc/mci_spec.h): parity first, then the regimes profiles/r05_spec.txt holds
set -u
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
