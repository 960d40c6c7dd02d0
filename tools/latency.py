#!/usr/bin/env python3
"""Development tool (GPU box): per-iteration wall time of the library loop (mci_integrate) at the reference's
typical sizes (neval 1e4..1e7): launch-bound regime.   latency.py --walks: the three refinement walks of train! side by side."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci

def walks():
    """the refinement walk of train!: prefix scan | the reference's recurrence (slots with the decisions given) | its general form"""
    import math
    L = math.sqrt(50.0)
    for name, mk in (("x2y2", lambda: mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), mci.catalog.x2y2())),
                     ("gauss16", lambda: mci.Engine(mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1), mci.catalog.gaussian(16)))):
        for neval in (10**4, 10**6, 10**8) if name == "gauss16" else (10**4, 10**6):
            for walk in ("scan", "serial", "serial_general"):
                eng = mk()
                eng.set_persistent("off")
                eng.set_train_walk(walk)
                n = 50 if neval < 10**8 else 12
                eng.integrate("vegas", neval=neval, niter=8, block=16, seed=1)
                best = 1e9
                for rep in range(3):
                    r = eng.integrate("vegas", neval=neval, niter=n, block=16, seed=1, first_iteration=8 + rep * n)
                    best = min(best, r["seconds"] / n * 1e6)
                print("%-8s neval=%-10d walk=%-15s %8.1f us/iteration   mean %.9f +- %.2e" % (name, neval, walk, best, r["mean"][0], r["stdev"][0]), flush=True)
                eng.close()


if __name__ == "__main__":
    if "--walks" in sys.argv:
        walks()
        sys.exit(0)
    for solver in ("vegas", "vegasmc", "mcmc"):
        for neval in (10**4, 10**5, 10**6, 10**7):
            cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
            eng = mci.Engine(cfg, mci.catalog.x2y2())
            # (light :vegas calls go persistent once that kernel's own translation unit exists: the automatic mode compiles it in the
            # background after 256 such calls of a process; here it is compiled on the spot and the automatic size rule applied by hand)
            if solver == "vegas" and neval * 2 < 2**19 and "--no-persistent" not in sys.argv:   # (a switch of this tool: the library reads no such variable)
                eng.set_persistent("on")
            eng.integrate(solver, neval=neval, niter=3, block=16, seed=1)
            persistent = solver == "vegas" and eng.last_integrate_persistent()
            t0 = time.perf_counter()
            r = eng.integrate(solver, neval=neval, niter=50, block=16, seed=1, first_iteration=3)
            dt = time.perf_counter() - t0
            wg = eng.kernel_times_ms(1)[1]
            if not persistent:
                eng.set_kernel_timing(1)   # the kernel's own duration: a second run with the HIP events on (they cost ~11 us per iteration)
                eng.integrate(solver, neval=neval, niter=50, block=16, seed=1, first_iteration=53)
                ms, wg, th = eng.kernel_times_ms(50)
            print("%-8s neval=%-9d  %8.1f us/iteration (library clock %8.1f)  %s  -> %8.1f Msamples/s   mean %.6f +- %.1e" % (
                solver, neval, dt / 50 * 1e6, r["seconds"] / 50 * 1e6,
                "one persistent launch, wg=%d" % wg if persistent else "kernel %8.1f us  wg=%d" % (float(np.median(ms)) * 1e3, wg),
                neval / (dt / 50) / 1e6, r["mean"][0], r["stdev"][0]), flush=True)
    # the headline integrand (16-D Gaussian on a shared 999-bin grid, :vegas) from launch-bound to throughput-bound sizes
    import math
    L = math.sqrt(50.0)
    eng = mci.Engine(mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1), mci.catalog.gaussian(16))
    eng.integrate("vegas", neval=10**8, niter=12, block=16, seed=1)
    it = 12
    for neval in (10**5, 10**6, 10**7, 3 * 10**7, 10**8):
        n = 50 if neval < 10**8 else 10
        eng.set_kernel_timing(0)
        eng.integrate("vegas", neval=neval, niter=3, block=16, seed=1, first_iteration=it, ignore=0)
        persistent = eng.last_integrate_persistent()   # (never for 16 draws per sample: the automatic rule stops at 7)
        t0 = time.perf_counter()
        eng.integrate("vegas", neval=neval, niter=n, block=16, seed=1, first_iteration=it + 3, ignore=0)
        dt = time.perf_counter() - t0
        ms, wg, th = eng.kernel_times_ms(1)
        if not persistent:
            eng.set_kernel_timing(1)
            eng.integrate("vegas", neval=neval, niter=n, block=16, seed=1, first_iteration=it + 3 + n, ignore=0)
            ms, wg, th = eng.kernel_times_ms(n)
        it += 3 + 2 * n
        print("C2 neval=%-10d %8.1f us/iteration  %s  -> %8.1f Msamples/s" % (
            neval, dt / n * 1e6, "one persistent launch, wg=%d th=%d" % (wg, th) if persistent else "kernel %8.1f us  wg=%d th=%d" % (float(np.median(ms)) * 1e3, wg, th),
            neval / (dt / n) / 1e6), flush=True)
