#!/bin/bash
# final r05 collection: bench line + rocprofv3 / PMC summaries of the same command; other configurations; the GPU suite on the final code
set -u
out=gpurun_out/r05_f
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python bench.py > $out/r05_bench_line.json 2> $out/bench.err
tail -c 600 $out/r05_bench_line.json; echo
bash profiles/collect.sh r05 bench > $out/collect_bench.log 2>&1
bash profiles/collect.sh r05_c3 c3 > $out/collect_c3.log 2>&1
bash profiles/collect.sh r05_c4 c4 > $out/collect_c4.log 2>&1
bash profiles/collect.sh r05_c5 c5 > $out/collect_c5.log 2>&1
bash profiles/collect.sh r05_bubble_mcmc bubble_mcmc > $out/collect_bubble.log 2>&1
bash profiles/collect.sh r05_default_call default_call > $out/collect_default.log 2>&1
cp profiles/r05*_kernel_stats.txt profiles/r05*_pmc_traffic.json $out/ 2>/dev/null
timeout 600 python tools/midsize_sweep.py gauss6 > $out/midsize_gauss6.txt 2>&1
timeout 900 python tools/bench_configs.py > $out/other_configs.txt 2>&1
du -sh gpurun_out
ls $out
