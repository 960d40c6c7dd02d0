# the several-lanes-per-chain units at -O2 (csrc/mci_jit.h): the failing case, the GPU suite, a fresh campaign, the timings
set -u
out=gpurun_out/r05_o2
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 200 python tools/repro_case.py 205 2>&1 | grep "^lanes" | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1; tail -3 $out/suite.txt
timeout 600 python tools/fuzz_layouts.py --carry --lanes 200 110 > $out/fuzz_a.txt 2>&1; tail -n 1 $out/fuzz_a.txt
timeout 300 python tools/spec_bench.py default 2>&1 | grep -v "warn\|resource"
timeout 300 python tools/mcmc_policy.py cold bubble 3e7 10 1 2>&1 | grep -v "warn\|resource" | tail -1
timeout 300 python tools/mcmc_policy.py cold cos 1e8 10 1 2>&1 | grep -v "warn\|resource" | tail -1
timeout 300 python tools/workload.py c5 --niter 10 --cold 2>&1 | grep -v "warn\|resource" | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
