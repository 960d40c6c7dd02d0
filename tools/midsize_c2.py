#!/usr/bin/env python3
"""Development tool (GPU box): the headline layout (16-D Gaussian on one shared 999-bin grid, :vegas) at the mid sizes the reference's own tests
run (test/montecarlo.jl:298-387): us per iteration of the library loop and the sample kernel's own duration, for the histogram-copy
count given (--copies N: csrc/mci_debug.h hist_copies) and whatever MCI_JIT_FLAGS selects (-DMCI_COPY_SUM_DPP=0, -DMCI_ZERO_B128=0).
usage: python tools/midsize_c2.py [--copies N] [label]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci
from mcintegration_jl_amd._lib import check, lib

args = sys.argv[1:]
copies = None
if "--copies" in args:
    copies = int(args[args.index("--copies") + 1])
    del args[args.index("--copies"):args.index("--copies") + 2]
    check(lib().mci_debug_override(b"hist_copies", copies, 1))
label = args[0] if args else "default"
L = math.sqrt(50.0)
eng = mci.Engine(mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1), mci.catalog.gaussian(16))
eng.integrate("vegas", neval=10**7, niter=12, block=16, seed=1)      # a trained map
it = 12
row = []
for neval in (10**5, 3 * 10**5, 10**6, 3 * 10**6, 10**7):
    n = 60
    eng.set_kernel_timing(0)
    eng.integrate("vegas", neval=neval, niter=3, block=16, seed=1, first_iteration=it, ignore=0)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        r = eng.integrate("vegas", neval=neval, niter=n, block=16, seed=1, first_iteration=it + 3 + rep * n, ignore=0)
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    eng.set_kernel_timing(1)
    eng.integrate("vegas", neval=neval, niter=n, block=16, seed=1, first_iteration=it + 3 + 3 * n, ignore=0)
    ms, wg, th = eng.kernel_times_ms(n)
    it += 3 + 4 * n
    row.append("%g: %.1f us (kernel %.1f, wg=%d th=%d)" % (neval, best, float(np.median(ms)) * 1e3, wg, th))
print("%-22s copies=%d  %s   mean %.7f +- %.1e" % (label, eng.histogram_copies(), " | ".join(row), r["mean"][0], r["stdev"][0]), flush=True)
mci.shutdown()
