#!/bin/bash
set -u
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
