set -u
out=gpurun_out/r05_log
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
BIAS_CARRY=off timeout 300 python tools/bias_ab.py full log vegasmc 64 1e7 10 16 4 > $out/log_carry_off.txt 2>&1
BIAS_NCHAIN=64 timeout 300 python tools/bias_ab.py full log vegasmc 64 1e7 10 16 4 > $out/log_nchain64.txt 2>&1
BIAS_NCHAIN=1 timeout 600 python tools/bias_ab.py full log vegasmc 32 1e7 10 16 4 > $out/log_nchain1.txt 2>&1
timeout 300 python tools/bias_ab.py full log vegas 64 1e7 10 16 4 > $out/log_vegas.txt 2>&1
timeout 300 python tools/bias_ab.py full log mcmc 64 1e7 10 16 4 > $out/log_mcmc.txt 2>&1
for f in $out/*.txt; do grep -v "resource_tracker\|warnings.warn" $f | head -16; done
