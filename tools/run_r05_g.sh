#!/bin/bash
# r05 evidence on the final code: randomised parity campaigns (with random lane groups / trees), the validation matrix, the GPU suite
set -u
out=gpurun_out/r05_g
mkdir -p $out
python -c "import torch" >/dev/null 2>&1
timeout 900 python tools/fuzz_layouts.py 0 150 > $out/fuzz_general.txt 2>&1
timeout 900 python tools/fuzz_layouts.py --lanes 0 150 > $out/fuzz_general_lanes.txt 2>&1
timeout 900 python tools/fuzz_layouts.py --carry 0 120 > $out/fuzz_carry.txt 2>&1
timeout 900 python tools/fuzz_layouts.py --carry --lanes 0 120 > $out/fuzz_carry_lanes.txt 2>&1
timeout 300 python tools/fuzz_layouts.py --persist 0 100 > $out/fuzz_persist.txt 2>&1
timeout 300 python tools/fuzz_layouts.py --pipe 0 40 > $out/fuzz_pipe.txt 2>&1
tail -n 1 $out/fuzz_*.txt
timeout 2400 python tools/validation_matrix.py 16 1e7 > $out/validation_matrix.txt 2>&1
cat $out/validation_matrix.txt
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.txt 2>&1
tail -5 $out/suite.txt
