#!/usr/bin/env python3
"""Development tool (GPU box): wall time of WHOLE integrate() calls at the reference's default size (neval=1e4, niter=10), the way a
parameter scan uses the reference: a new Configuration per call (same integrand source: the code object comes from the kernel
cache), and the same Configuration reused.  Where the time of a call goes (create / module load / iterations / read-back)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci

if __name__ == "__main__":
    src = "return x[0] * x[0] + x[1] * x[1];"
    for solver in ("vegas", "vegasmc", "mcmc"):
        mci.integrate(src, var=mci.Continuous(0.0, 1.0), dof=[[2]], solver=solver, neval=1e4)          # JIT / cache load
        n = 30
        t0 = time.perf_counter()
        for i in range(n):
            r = mci.integrate(src, var=mci.Continuous(0.0, 1.0), dof=[[2]], solver=solver, neval=1e4)
        fresh = (time.perf_counter() - t0) / n
        cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
        mci.integrate(src, config=cfg, solver=solver, neval=1e4)
        t0 = time.perf_counter()
        for i in range(n):
            r = mci.integrate(src, config=cfg, solver=solver, neval=1e4)
        reuse = (time.perf_counter() - t0) / n
        print("%-8s integrate(neval=1e4, niter=10): new Configuration per call %8.3f ms, same Configuration %8.3f ms   (%s)" % (
            solver, fresh * 1e3, reuse * 1e3, r), flush=True)
    # steady state: the same Configuration over and over, persistent launch off / on (the automatic mode compiles that kernel in the
    # background once a process has made 256 launch-bound calls of it; "on" compiles it on the spot)
    import numpy as np
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
    eng = cfg._engine
    for mode in ("off", "on"):
        eng.set_persistent(mode)
        mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
        ts = []
        for i in range(300):
            t0 = time.perf_counter()
            r = mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        lib = []
        for i in range(300):
            lib.append(eng.integrate("vegas", neval=10000, niter=10, block=16, seed=1)["seconds"])
        print("vegas    steady state, persistent launch %-3s: integrate(neval=1e4, niter=10) median %.3f ms (min %.3f, p90 %.3f); library clock %.3f ms" % (
            mode, np.median(ts), ts.min(), np.percentile(ts, 90), np.median(lib) * 1e3), flush=True)
    eng.set_persistent("auto")
    # profile of calls on a reused Configuration, then of fresh ones
    import cProfile, pstats
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(50):
        mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(10):
        mci.integrate(src, var=mci.Continuous(0.0, 1.0), dof=[[2]], solver="vegas", neval=1e4)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
