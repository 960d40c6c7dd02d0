#!/bin/bash
set -u
out=gpurun_out/r05_spec2
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 2400 python -m pytest tests/test_hip_spec.py -q -p no:cacheprovider > $out/pytest_spec.txt 2>&1
tail -40 $out/pytest_spec.txt
# the :vegasmc bias of cold calls at 1e8: is it the carried chains?
for cs in "c5" "bubble"; do
  BIAS_LANES=1 timeout 600 python tools/bias_ab.py full $cs vegasmc 64 1e8 10 16 4 > $out/full_${cs}_vegasmc_carry_auto.txt 2>&1
  BIAS_LANES=1 BIAS_CARRY=off timeout 900 python tools/bias_ab.py full $cs vegasmc 64 1e8 10 16 4 > $out/full_${cs}_vegasmc_carry_off.txt 2>&1
  BIAS_LANES=1 BIAS_CARRY=off BIAS_NCHAIN=256 timeout 900 python tools/bias_ab.py full $cs vegasmc 64 1e8 10 16 4 > $out/full_${cs}_vegasmc_carry_off_nchain256.txt 2>&1
  BIAS_LANES=1 BIAS_NCHAIN=256 timeout 900 python tools/bias_ab.py full $cs vegasmc 64 1e8 10 16 4 > $out/full_${cs}_vegasmc_carry_auto_nchain256.txt 2>&1
done
tail -n +1 $out/full_*.txt
