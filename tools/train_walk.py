#!/usr/bin/env python3
"""Development tool (GPU box): cost of the refinement walk of train! -- per-iteration time of a launch-bound :vegas loop
(neval = 1e4) with the reference's serial recurrence against the prefix-scan form, for 1 leaf and for 32 leaves."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcintegration_jl_amd as mci

if __name__ == "__main__":
    for label, cfg_f, f in (("1 leaf x 999 bins", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), mci.catalog.x2y2),
                            ("32 leaves x 999 bins", lambda: mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]]), lambda: mci.catalog.genz_product_peak(32)),
                            ("1 leaf x 3999 bins", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0, ninc=4000), dof=[[2]]), mci.catalog.x2y2)):
        for walk in ("scan", "serial"):
            eng = mci.Engine(cfg_f(), f())
            eng.set_train_walk(walk)
            eng.integrate("vegas", neval=10**4, niter=5, block=16, seed=1)
            t0 = time.perf_counter()
            r = eng.integrate("vegas", neval=10**4, niter=200, block=16, seed=1, first_iteration=5)
            dt = time.perf_counter() - t0
            print("%-22s walk=%-7s %8.1f us/iteration   mean %.8f +- %.1e" % (label, walk, dt / 200 * 1e6, r["mean"][0], r["stdev"][0]), flush=True)
