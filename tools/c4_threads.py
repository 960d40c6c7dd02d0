#!/usr/bin/env python3
"""Development tool (GPU box): workgroup size x compile-time variant of the C4 sample pass (per-kernel HIP-event times)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from c4_sweep import run
for flags in sys.argv[1:] or [""]:
    for T in (512, 768, 1024):
        print("threads=%d flags=%-40s %s" % (T, flags, run(T, 0, None, flags, 32)), flush=True)
