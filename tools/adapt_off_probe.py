#!/usr/bin/env python3
"""Development tool (GPU box): a chain solver with adapt = false (every iteration counted, every launch on the map as it is) over seeds:
deviation of the estimate in its reported errors.   usage: python tools/adapt_off_probe.py <case> <solver> [nseeds] [neval] [niter] [nchain]   (nchain = 1: the reference's own chain per block)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mcintegration_jl_amd as mci
from mcmc_policy import case

if __name__ == "__main__":
    name, solver = sys.argv[1], sys.argv[2]
    nseeds = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    neval = int(float(sys.argv[4])) if len(sys.argv) > 4 else 10**7
    niter = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    nchain = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    devs, its, t0 = [], [], time.perf_counter()
    for seed in range(1, nseeds + 1):
        cfg, f, meas, exact = case(name, seed=seed)
        eng = mci.Engine(cfg, f, measure=meas)
        r = eng.integrate(solver, neval=neval, niter=niter, block=16, seed=seed, adapt=False, ignore=0, nchain=nchain)
        ex = np.asarray(exact, dtype=float).ravel()
        devs.append((np.asarray(r["mean"]).ravel() - ex) / np.asarray(r["stdev"]).ravel())
        its.append(np.asarray(r["iter_mean"]).reshape(niter, -1) - ex)
        carried = eng.last_chain_launch()
        eng.close()
    devs, its = np.array(devs), np.array(its)
    print("%s :%s adapt=false ignore=0 nchain=%s  %d seeds x integrate(neval=%.0e, niter=%d)  %.2f s per run; last launch (chains per block, carried) = %s" % (
        name, solver, nchain or "auto", nseeds, neval, niter, (time.perf_counter() - t0) / nseeds, carried))
    print("  mean deviation per run in units of one run's error :", np.round(devs.mean(0), 2), " rms", np.round(np.sqrt((devs**2).mean(0)), 2))
    for i in range(niter):
        m, s = its[:, i].mean(0), its[:, i].std(0, ddof=1) / np.sqrt(nseeds)
        print("  iteration %2d: (mean over seeds - exact) / its error  %s" % (i + 1, np.round(m / s, 2)))
    mci.shutdown()
