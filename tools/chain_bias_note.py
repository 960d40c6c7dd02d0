#!/usr/bin/env python3
"""Development tool (GPU box): the estimate behind report(result)'s "one chain per block" note (statistics.chain_estimator_bias) against the
measured bias -- mean deviation per run over seeds in units of the reported error -- for calls with many short blocks.
usage: python tools/chain_bias_note.py [nseeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mcintegration_jl_amd as mci
from mcmc_policy import case

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print("%-8s %-8s %6s %8s | %-30s | predicted z  acceptance  tau   times = steps per block / tau" % ("case", "solver", "block", "neval", "measured mean dev per run"))
for name in ("c5", "sphere2", "x2", "log"):
    for solver in ("mcmc", "vegasmc"):
        for block, neval in ((16, 1e6), (64, 1e6), (256, 1e6), (16, 1e4), (64, 1e5)):
            devs, zs, b = [], [], None
            for seed in range(1, n + 1):
                cfg, f, meas, exact = case(name, seed=seed)
                res = mci.integrate(f, config=cfg, solver=solver, measure=meas, neval=neval, niter=10, block=block)
                devs.append((np.asarray(res._flat_mean).ravel() - np.asarray(exact, dtype=float).ravel()) / np.asarray(res._flat_std).ravel())
                b = res.chain_bias
                zs.append(b["z"] if b else float("nan"))
                cfg._engine.close()
            devs = np.array(devs)
            print("%-8s %-8s %6d %8.0e | %-30s | %8.2f %10s %6s %8s" % (name, solver, block, neval, np.round(devs.mean(0), 2), np.nanmean(zs),
                  "%.2f" % b["acceptance"] if b else "-", "%.1f" % b["tau"] if b else "-", "%.0f" % b["times"] if b else "(several chains per block)"), flush=True)
mci.shutdown()
