# automatic :vegasmc chains started afresh (iterations 1 and 2 of a cold call): their length, in burn-in floors, and the part of them that is not
# measured against the bias of the second iteration and the time per run (profiles/r05_bias.txt A5; csrc/mci_debug.h fresh_floors, fresh_burnin_pct)
set -u
out=gpurun_out/r05_floors
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
for fl in 8 16 32 64; do
  BIAS_FRESH_FLOORS=$fl timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_fl$fl.txt 2>&1
done
for fl in 8 32 64; do
  BIAS_FRESH_FLOORS=$fl timeout 400 python tools/bias_ab.py full bubble vegasmc 16 1e8 10 16 2 > $out/bubble_fl$fl.txt 2>&1
  BIAS_FRESH_FLOORS=$fl timeout 400 python tools/bias_ab.py full c5 vegasmc 16 1e8 10 16 2 > $out/c5_fl$fl.txt 2>&1
  BIAS_FRESH_FLOORS=$fl timeout 300 python tools/bias_ab.py full cos vegasmc 32 1e7 10 16 4 > $out/cos_fl$fl.txt 2>&1
done
for b in 25 50 75; do
  BIAS_FRESH_BURNIN=$b timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_burn$b.txt 2>&1
done
BIAS_FRESH_FLOORS=16 BIAS_FRESH_BURNIN=50 timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_fl16_burn50.txt 2>&1
BIAS_FRESH_FLOORS=32 BIAS_FRESH_BURNIN=50 timeout 300 python tools/bias_ab.py full log vegasmc 128 1e7 10 16 4 > $out/log_fl32_burn50.txt 2>&1
for f in $out/*.txt; do echo "== $f"; grep -v "resource_tracker\|warnings.warn" $f | head -9; done
