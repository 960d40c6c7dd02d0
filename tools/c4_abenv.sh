#!/bin/bash
# Development tool (GPU box): A/B of environment-level variants of the C4 sample pass (tools/ab_c2.py c4), each run twice
for e in "MCI_L1_PHASE=1 MCI_THREADS=512" "MCI_L1_PHASE=0 MCI_THREADS=1024" "MCI_L1_PHASE=0 MCI_THREADS=768" "MCI_L1_PHASE=1 MCI_THREADS=512" "MCI_L1_PHASE=0 MCI_THREADS=1024"; do
  echo "env $e"; env $e python tools/ab_c2.py c4 "" 2>&1 | tail -1 | cut -c1-200
done
