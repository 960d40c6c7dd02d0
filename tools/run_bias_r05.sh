#!/bin/bash
# GPU box: the three bias flags of VERDICT r04 at their own configurations (tools/bias_ab.py) -> gpurun_out/r05_bias/
set -u
out=gpurun_out/r05_bias
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
NS=${NS:-256}
# (a) cold integrate(neval = 1e8, niter = 10) of the chain solvers' BASELINE configurations, iteration by iteration
for cs in "bubble vegasmc" "c5 vegasmc" "c5 mcmc"; do
  set -- $cs
  timeout 900 python tools/bias_ab.py full $1 $2 ${NS_FULL:-128} 1e8 10 16 4 > $out/full_$1_$2_1e8.txt 2>&1
done
# (b) the reference's chain against the automatic chains at the same (neval = 1e6, block)
for cs in "c5 vegasmc" "c5 mcmc" "sphere2 vegasmc" "sphere2 mcmc" "bubble vegasmc" "bubble mcmc"; do
  set -- $cs
  for blk in 16 64; do
    timeout 1500 python tools/bias_ab.py ab $1 $2 $NS 1e6 10 $blk 8 > $out/ab_$1_$2_b$blk.txt 2>&1
  done
done
tail -n +1 $out/*.txt
