#!/usr/bin/env python3
"""Development tool (GPU box): one :mcmc configuration for rocprofv3 --pmc (instructions per chain step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcintegration_jl_amd as mci
nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss())
eng.integrate("mcmc", neval=10**8, niter=3, block=16, seed=1, nchain=nchain)
r = eng.integrate("mcmc", neval=10**8, niter=4, block=16, seed=1, first_iteration=3, nchain=nchain)
print("chains per block", nchain or "automatic", "s/iter", r["seconds"] / 4)
