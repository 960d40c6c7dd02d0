#!/usr/bin/env python3
"""Development tool (GPU box): C3 bubble :vegasmc at 1e8 for rocprofv3 --pmc (instructions per chain step)."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcintegration_jl_amd as mci
PI = math.pi
p = mci.catalog.bubble_parameters()
var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
       mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
eng = mci.Engine(mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)]), mci.catalog.bubble(), measure=mci.bin_by(4))
r = eng.integrate("vegasmc", neval=10**8, niter=6, block=16, seed=1)
ms, wg, th = eng.kernel_times_ms(6)
print("kernel ms", np.median(ms), wg, th)
