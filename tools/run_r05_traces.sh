#!/bin/bash
set -u
out=gpurun_out/r05_c
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_battery.py -q -p no:cacheprovider -k "closure or bias or unbiased or cold_vegasmc" > $out/pytest_a.txt 2>&1
tail -30 $out/pytest_a.txt
timeout 300 python tools/mcmc_policy.py trace c5 1e8 10 > $out/trace_c5_auto.txt 2>&1
POLICY_LANES=1 timeout 300 python tools/mcmc_policy.py trace c5 1e8 10 > $out/trace_c5_lanes1.txt 2>&1
timeout 300 python tools/mcmc_policy.py trace bubble 3e7 10 > $out/trace_bubble_auto.txt 2>&1
timeout 300 python tools/mcmc_policy.py trace cos 1e8 10 > $out/trace_cos_auto.txt 2>&1
for cs in c5 bubble; do
  timeout 600 python tools/bias_ab.py full $cs vegasmc 64 1e8 10 16 4 > $out/full_${cs}_vegasmc_resampled.txt 2>&1
done
timeout 600 python tools/bias_ab.py full c5 vegasmc 64 1e6 10 16 4 > $out/full_c5_vegasmc_1e6_resampled.txt 2>&1
timeout 900 python bench.py --min-seconds 3 > $out/bench.json 2> $out/bench.err
tail -n +1 $out/trace_*.txt $out/full_*.txt; python -c "
import json;j=json.load(open('$out/bench.json'));r=j['roofline'];print(j['value'],j['ms_per_step'],r['frac'],r.get('frac_self_calibrated'),r['clock'],r['valu_datasheet'].get('issue_ns_per_wave_sample'),r['kernel_ms_avg'])"
