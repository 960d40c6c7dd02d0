#!/usr/bin/env python3
"""A Python closure as integrand, four ways (needs an MI355X):
   traced into device source (the closure runs once on symbolic draws, the kernels are JIT-compiled around what it computes),
   as a host batch callback (the closure is called with every batch of draws: PCIe- and host-bound),
   in the reference's in-place form -- integrate(f; inplace = true), f(var, weights, config) stores its weights (main.jl:26) --,
   and the same function written as a HIP C++ body by hand.
Reference call:  integrate((x, c) -> exp(-(x[1]^2 + x[2]^2 + x[3]^2) / 2) / (2pi)^1.5; var = Continuous(-5, 5), dof = [[3]], solver = :vegas)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcintegration_jl_amd as mci

f = lambda x, c: np.exp(-np.sum(x * x, axis=0) / 2) / (2 * np.pi) ** 1.5
body = "return exp(-(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) / 2) / 15.749609945722419;"
print(mci.trace_integrand(f, mci.Configuration(var=mci.Continuous(-5.0, 5.0), dof=[[3]])).body)


def f_inplace(x, weights, c):
    weights[0] = np.exp(-np.sum(x * x, axis=0) / 2) / (2 * np.pi) ** 1.5


for name, integrand, kw in (("traced closure", f, {}), ("host closure", f, dict(trace=False)), ("in-place closure", f_inplace, dict(inplace=True)),
                            ("in-place, host", f_inplace, dict(inplace=True, trace=False)), ("device source", body, {})):   # (tracing is the default)
    t0 = time.time()
    r = mci.integrate(integrand, var=mci.Continuous(-5.0, 5.0), dof=[[3]], solver="vegas", neval=1e6, niter=10, seed=1, print=-1, **kw)
    print("%-17s %.6f +- %.1e   %.2f s" % (name, r.mean[0], r.stdev[0], time.time() - t0))
