#!/usr/bin/env python3
"""Development tool (GPU box): 200 vegas iterations at neval=1e5 (launch-bound regime) for rocprofv3 --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcintegration_jl_amd as mci
neval = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**5
eng = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), mci.catalog.x2y2())
eng.integrate("vegas", neval=neval, niter=3, block=16, seed=1)
r = eng.integrate("vegas", neval=neval, niter=200, block=16, seed=1, first_iteration=3)
print("us/iteration", r["seconds"] / 200 * 1e6)
