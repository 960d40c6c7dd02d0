#!/usr/bin/env python3
"""Development tool (GPU box): a Python closure traced into device source (integrate(..., trace=True)) next to the same closure on the
host callback path, and a two-integrand closure with a select under the three solvers."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import mcintegration_jl_amd as mci
f = lambda x, c: np.exp(-np.sum(x * x, axis=0) / 2) / (2 * np.pi) ** (len(x) / 2)
for trace in (True, False):
    t0 = time.time()
    r = mci.integrate(f, var=mci.Continuous(-5.0, 5.0), dof=[[4]], solver="vegas", neval=1000000, niter=10, seed=7, trace=trace, print=-1)
    print("trace=%s: %.6f +- %.2e  chi2 %.2f   %.2f s" % (trace, r.mean[0], r.stdev[0], r.chi2[0], time.time() - t0), flush=True)
g = lambda x, c: (x[0] ** 2 + x[1] ** 2, mci.trace.where(x[0] > 0.5, x[1], 0.0))
for solver in ("vegas", "vegasmc", "mcmc"):
    r = mci.integrate(g, var=mci.Continuous(0.0, 1.0), dof=[[2], [2]], solver=solver, neval=400000, niter=10, seed=7, trace=True, print=-1)
    print("%-8s %.5f +- %.1e (2/3)   %.5f +- %.1e (1/4)" % (solver, r.mean[0], r.stdev[0], r.mean[1], r.stdev[1]), flush=True)
