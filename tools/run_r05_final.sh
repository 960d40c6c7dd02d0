#!/bin/bash
# the last GPU batch of round 5, on the final code objects: GPU suite, smoke, bench line + rocprofv3 / PMC of the same command, C3 / C5 :vegasmc
# and the default call re-profiled (their kernels changed late), every configuration cold and trained
set -u
out=gpurun_out/r05_final
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -4 $out/suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python bench.py > $out/r05_bench_line.json 2> $out/bench.err
bash profiles/collect.sh r05 bench > $out/collect_bench.log 2>&1
bash profiles/collect.sh r05_c3 c3 > $out/collect_c3.log 2>&1
bash profiles/collect.sh r05_default_call default_call > $out/collect_default.log 2>&1
cp profiles/r05_kernel_stats.txt profiles/r05_pmc_traffic.json profiles/r05_c3_kernel_stats.txt profiles/r05_c3_pmc_traffic.json profiles/r05_default_call_kernel_stats.txt profiles/r05_default_call_pmc_traffic.json $out/
timeout 900 python tools/bench_configs.py > $out/other_configs.txt 2>&1
timeout 300 python tools/spec_bench.py default > $out/default.txt 2>&1
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05_final/r05_bench_line.json")); r=j["roofline"]
print(j["value"], j["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["frac_self_calibrated"], r["clock"]["sclk_mhz_avg"], r["traffic"], j["config"]["code_object"])
print(json.load(open("gpurun_out/r05_final/r05_pmc_traffic.json"))["code_object"])
PY
cat $out/default.txt
