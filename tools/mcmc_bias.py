#!/usr/bin/env python3
"""Development tool (GPU box): high-statistics bias check of the many-chain :mcmc decomposition on C5 (4 nested Gaussians on a
12-D pool), automatic chain length, several seeds: 5 training + 20 production iterations x 1e8 steps each."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci

ex5 = np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nchain = int(sys.argv[2]) if len(sys.argv) > 2 else 0
devs, means = [], []
for seed in range(1, nseeds + 1):
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]])
    eng = mci.Engine(cfg, mci.catalog.nested_gauss())
    eng.integrate("mcmc", neval=10**8, niter=5, block=16, seed=seed, nchain=nchain)
    r = eng.integrate("mcmc", neval=10**8, niter=20, block=16, seed=seed, first_iteration=5, ignore=0, nchain=nchain)
    hh = eng.hold_histogram()
    print("seed %d  chains/launch %d  %.2f s  mean-exact=%s  dev=%s  chi2=%s" % (seed, int(hh.sum()), r["seconds"], np.array2string(r["mean"] - ex5, precision=2),
          np.round((r["mean"] - ex5) / r["stdev"], 2), np.round(r["chi2"], 2)), flush=True)
    devs.append((r["mean"] - ex5) / r["stdev"]); means.append(r["mean"] - ex5)
means = np.array(means)
print("mean over seeds of (mean-exact): %s +- %s" % (np.array2string(means.mean(0), precision=2), np.array2string(means.std(0, ddof=1) / math.sqrt(len(means)), precision=2)))
