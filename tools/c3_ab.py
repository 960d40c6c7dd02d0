#!/usr/bin/env python3
"""Development tool (GPU box): A/B of JIT flags on the C3 bubble :vegas kernel."""
import json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, math
sys.path.insert(0, %r)
import numpy as np
import mcintegration_jl_amd as mci
PI = math.pi
p = mci.catalog.bubble_parameters()
var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
       mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
cfg = mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=1)
eng = mci.Engine(cfg, mci.catalog.bubble(), measure=mci.bin_by(4))
eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=1)
eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=2, first_iteration=5)
ms, wg, th = eng.kernel_times_ms(5)
print(json.dumps(dict(ms=float(np.median(ms)), wg=wg, threads=th)))
''' % ROOT
for flags in [""] + sys.argv[1:]:
    env = dict(os.environ); env["MCI_KERNEL_CACHE"] = "/tmp/mci_c3_cache"
    if flags: env["MCI_JIT_FLAGS"] = flags
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("flags=%-20s %s" % (flags, out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-300:]), flush=True)
