#!/usr/bin/env python3
"""PROTOTYPE measurement (GPU box): the packed 52-bit :vegas stream (three draws in five Philox words) against the word-pair stream in the
pipelined headline loop, interleaved runs of `tools/workload.py c2` in fresh processes (MCI_JIT_FLAGS=-DMCI_PROTO_PACK=1 switches the loop)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(3):
    for flags in ("", "-DMCI_PROTO_PACK=1"):
        env = dict(os.environ, MCI_JIT_FLAGS=flags)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "workload.py"), "c2", "--niter", "40"], env=env, capture_output=True, text=True)
        print("%-22s %s" % (flags or "(word pairs)", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
