# the several-lanes-per-chain units at -O3 without si-optimize-exec-masking-pre-ra (csrc/mci_jit.h)
set -u
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
