#!/bin/bash
# Development tool (GPU box): histogram copies x workgroup size on a BASELINE workload (tools/ab_c2.py timing).
#   usage: tools/hcopy_sweep.sh [c2|c5v|c2i] "copies..." "threads..."
W=${1:-c2}
for th in ${3:-256 512 1024}; do
  for hc in ${2:-1 2 4 8 16}; do
    echo -n "threads=$th copies=$hc  "
    MCI_THREADS=$th MCI_HIST_COPIES=$hc python tools/ab_c2.py $W "" 2>&1 | tail -1
  done
done
