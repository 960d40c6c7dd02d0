#!/usr/bin/env python3
"""Development tool (GPU box): many-chain bias check on integrands with unbounded |f|/q (sticky states)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci
PI = math.pi
cases = [("log/sqrt", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]), mci.catalog.log_over_sqrt(), [-4.0]),
         ("singular2", lambda: mci.Configuration(var=mci.Continuous(0.0, PI), dof=[[3]]), mci.catalog.singular2(), [1.3932039296856769]),
         ("sphere2", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]]), mci.catalog.sphere2(), [PI / 4, PI / 6])]
for name, mk, f, exact in cases:
    for solver, nchain in ([(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]] or [("vegas", 0), ("vegasmc", 0), ("vegasmc", 16384), ("mcmc", 0), ("mcmc", 4096)]):
        eng = mci.Engine(mk(), f)
        eng.integrate(solver, neval=10**8, niter=5, block=16, seed=1, nchain=nchain)
        r = eng.integrate(solver, neval=10**8, niter=20, block=64, seed=1, first_iteration=5, ignore=0, nchain=nchain)
        print("%-10s %-8s nchain=%-6d (last launch: %d chains) %.2f s mean=%s sigma=%s dev=%s chi2=%s" % (name, solver, nchain, int(eng.hold_histogram().sum()) if solver == "mcmc" else -1, r["seconds"], np.array2string(r["mean"], precision=8),
              np.array2string(r["stdev"], precision=2), np.array2string((r["mean"] - np.array(exact)) / r["stdev"], precision=2),
              np.array2string(r["chi2"], precision=2)), flush=True)
