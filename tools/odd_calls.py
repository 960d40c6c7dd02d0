#!/usr/bin/env python3
"""Development tool (GPU box): calls a user of the reference makes that are not the default one -- a Configuration continued by a second
integrate(), very small and very many blocks, ignore = 0, niter = 2, measurefreq > 1, a reweight_goal -- over seeds, chain solvers and
:vegas: mean deviation per run in units of the reported error and its rms (1 expected).   usage: python tools/odd_calls.py [nseeds]"""
import math
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


def run(label, name, solver, nseeds, calls):
    """calls: list of keyword dicts, one integrate() each on the SAME configuration; the last result is judged"""
    devs, t0 = [], time.perf_counter()
    for seed in range(1, nseeds + 1):
        cfg, f, meas, exact = case(name, seed=seed)
        res = None
        for kw in calls:
            res = mci.integrate(f, config=cfg, solver=solver, measure=meas, **kw)
        ex = np.asarray(exact, dtype=float).ravel()
        devs.append((np.asarray(res._flat_mean).ravel() - ex) / np.asarray(res._flat_std).ravel())
        cfg._engine.close() if getattr(cfg, "_engine", None) is not None else None
    devs = np.array(devs)
    print("%-34s %-7s :%-8s mean dev per run %s  rms %s   (%.2f s per run)" % (
        label, name, solver, np.round(devs.mean(0), 2), np.round(np.sqrt((devs ** 2).mean(0)), 2), (time.perf_counter() - t0) / nseeds), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for solver in ("vegasmc", "mcmc", "vegas"):
        for name in ("c5", "sphere2"):
            run("continued config 5 + 5 iterations", name, solver, n, [dict(neval=1e6, niter=5), dict(neval=1e6, niter=5)])
            run("continued, second call adapt=false", name, solver, n, [dict(neval=1e6, niter=5), dict(neval=1e6, niter=5, adapt=False)])
            run("neval=1e3 (62 per block)", name, solver, n, [dict(neval=1e3, niter=10)])
            run("block=256 neval=1e6", name, solver, n, [dict(neval=1e6, niter=10, block=256)])
            run("ignore=0", name, solver, n, [dict(neval=1e6, niter=5, ignore=0)])
            run("niter=2", name, solver, n, [dict(neval=1e6, niter=2)])
            run("measurefreq=5", name, solver, n, [dict(neval=1e6, niter=6, measurefreq=5)])
        run("reweight_goal", "sphere2", solver, n, [dict(neval=1e6, niter=6, reweight_goal=[1.0, 3.0, 2.0])])
    mci.shutdown()
