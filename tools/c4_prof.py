#!/usr/bin/env python3
"""Development tool (GPU box): the C4 workload for rocprofv3 --stats (sample pass vs histogram replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcintegration_jl_amd as mci
D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * D), dof=[[1]], seed=1)
T = int(os.environ.get("MCI_C4_THREADS", "0"))
eng = mci.Engine(cfg, mci.catalog.genz_product_peak(D), **(dict(threads=T) if T else {}))
eng.integrate("vegas", neval=10**8, niter=6, block=16, seed=1)
