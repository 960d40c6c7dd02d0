#!/usr/bin/env python3
"""Development tool (GPU box): high-statistics bias check of the many-chain MCMC decomposition (C5 family)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci

ex5 = np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
for nchain in (0, 256, 16384):
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]])
    eng = mci.Engine(cfg, mci.catalog.nested_gauss())
    eng.integrate("mcmc", neval=10**8, niter=5, block=16, seed=1, nchain=nchain)
    r = eng.integrate("mcmc", neval=10**8, niter=40, block=64, seed=1, first_iteration=5, ignore=0, nchain=nchain)
    print("nchain=%-6d mean=%s sigma=%s dev=%s chi2=%s" % (nchain, r["mean"], r["stdev"], (r["mean"] - ex5) / r["stdev"], r["chi2"]), flush=True)
