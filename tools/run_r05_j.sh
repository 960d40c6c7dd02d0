#!/bin/bash
# final code: GPU suite, bench line, latency tables, the default-call profile again (its kernel changed), the bubble A/B of profiles/r05_bias.txt B
set -u
out=gpurun_out/r05_j
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -5 $out/suite.txt
timeout 600 python bench.py > $out/bench_line.json 2> $out/bench.err
MCI_BENCH_FORCE_COMM=1 timeout 600 python bench.py --min-seconds 2 --no-cpu-baseline > $out/bench_line_forced_comm.json 2> $out/bench_fc.err
timeout 600 python tools/latency.py > $out/latency.txt 2>&1
bash profiles/collect.sh r05_default_call default_call > $out/collect_default.log 2>&1
cp profiles/r05_default_call_kernel_stats.txt profiles/r05_default_call_pmc_traffic.json $out/
timeout 400 python tools/bias_ab.py ab bubble vegasmc 32 1e6 10 16 1 > $out/ab_bubble_vegasmc_b16.txt 2>&1
timeout 600 python tools/bias_ab.py ab bubble mcmc 32 1e6 10 16 1 > $out/ab_bubble_mcmc_b16.txt 2>&1
tail -n +1 $out/latency.txt $out/ab_*.txt
python - <<'PY'
import json
for f in ("gpurun_out/r05_j/bench_line.json","gpurun_out/r05_j/bench_line_forced_comm.json"):
    j=json.load(open(f)); r=j["roofline"]; print(f, j["value"], j["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["clock"]["sclk_mhz_avg"], j["comm"]["kind"], j["comm"].get("collectives"))
PY
