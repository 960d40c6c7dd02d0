#!/usr/bin/env python3
"""Development tool (GPU box): does the many-chain :mcmc estimate of the bubble depend on chain length / burn-in?"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci
PI = math.pi
p = mci.catalog.bubble_parameters()

def bub():
    var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
           mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
    return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])

ref = None
import json
CASES = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [["vegas", 0, 0.1], ["vegasmc", 0, 0.1], ["mcmc", 64, 0.1], ["mcmc", 1024, 0.1], ["mcmc", 8192, 0.1], ["mcmc", 8192, 8.0]]
for solver, nchain, tr in CASES:
    eng = mci.Engine(bub(), mci.catalog.bubble(), measure=mci.bin_by(4))
    eng.integrate(solver, neval=10**8, niter=5, block=16, seed=1, nchain=nchain, thermal_ratio=tr)
    r = eng.integrate(solver, neval=10**8, niter=20, block=64, seed=1, first_iteration=5, ignore=0, nchain=nchain, thermal_ratio=tr)
    if ref is None:
        ref = r["mean"].copy()
    print("%-8s nchain=%-6d thermal_ratio=%-4g mean=%s sigma=%s  (mean-vegas)/sigma=%s" % (
        solver, nchain, tr, np.array2string(r["mean"], precision=7), np.array2string(r["stdev"], precision=2),
        np.array2string((r["mean"] - ref) / r["stdev"], precision=2)), flush=True)
