set -u
out=gpurun_out/r05_a4
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
for cs in x2 log sphere2 hyper cos c5 bubble; do
  timeout 300 python tools/bias_ab.py full $cs vegasmc 64 1e7 10 16 4 > $out/full_${cs}_vegasmc_1e7.txt 2>&1
done
for f in $out/full_*; do grep -v "resource_tracker\|warnings.warn" $f | head -4; grep "iteration  2 \|iteration  3 " $f; done
