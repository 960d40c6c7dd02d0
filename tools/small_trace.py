#!/usr/bin/env python3
"""Development tool (GPU box, under rocprofv3 --kernel-trace): 60 :vegas iterations of a 2-D integrand at neval = 1e4, for the
timeline of the launch-bound regime (kernel durations and the gaps between them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcintegration_jl_amd as mci
neval = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**4
cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
eng = mci.Engine(cfg, mci.catalog.x2y2())
eng.integrate("vegas", neval=neval, niter=3, block=16, seed=1)
eng.integrate("vegas", neval=neval, niter=60, block=16, seed=1, first_iteration=3)
