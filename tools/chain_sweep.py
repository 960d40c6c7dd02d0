#!/usr/bin/env python3
"""Development tool (GPU box): chains-per-block sweep for the chain solvers (throughput and bias against the exact value)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mcintegration_jl_amd as mci
from catalog_params import bubble_exact

PI = math.pi


def run(name, mk, f, solver, neval, exact, nchain, measure=None, threads=None):
    cfg = mk()
    eng = mci.Engine(cfg, f, measure=measure, threads=threads)
    eng.compile(solver)
    eng.integrate(solver, neval=neval, niter=5, block=16, seed=1, nchain=nchain)
    r = eng.integrate(solver, neval=neval, niter=5, block=16, seed=1, first_iteration=5, ignore=0, nchain=nchain)
    ms, wg, th = eng.kernel_times_ms(5)
    dev = (r["mean"] - np.atleast_1d(exact)) / r["stdev"]
    print("%-18s %-8s nchain=%-6d %9.1f Msteps/s kernel %8.3f ms (wg=%d,th=%d) sigma=%s dev=%s" % (
        name, solver, nchain, neval * 5 / r["seconds"] / 1e6, float(np.median(ms)), wg, th,
        np.array2string(r["stdev"], precision=2), np.array2string(dev, precision=2)), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c5", "c3"]
    p = mci.catalog.bubble_parameters()

    def bub():
        var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
               mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
        return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])

    def c5():
        return mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]])
    ex5 = [math.erf(5.0) ** d for d in (3, 6, 9, 12)]
    for nchain in (0, 1024, 2048, 4096, 8192, 16384, 32768):
        if "c5" in which:
            run("C5 nested gauss", c5, mci.catalog.nested_gauss(), "mcmc", 10**8, ex5, nchain)
        if "c5mc" in which:
            run("C5 nested gauss", c5, mci.catalog.nested_gauss(), "vegasmc", 10**8, ex5, nchain)
        if "c3" in which:
            run("C3 bubble", bub, mci.catalog.bubble(), "vegasmc", 10**8, bubble_exact(), nchain, measure=mci.bin_by(4))
        if "c3m" in which:
            run("C3 bubble", bub, mci.catalog.bubble(), "mcmc", 10**8, bubble_exact(), nchain, measure=mci.bin_by(4))
