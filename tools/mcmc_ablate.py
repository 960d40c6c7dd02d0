#!/usr/bin/env python3
"""Compile-time ablation of mci_mcmc_chains on C5 (841 chains per block = the automatic setting): what each part of a chain step costs.
Run as:  MCI_JIT_FLAGS="-DMCI_ABL_..." python tools/mcmc_ablate.py   (flags: MCI_ABL_NOHOLD, MCI_ABL_NOMCHIST)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcintegration_jl_amd as mci
eng = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss())
eng.integrate("mcmc", neval=10**8, niter=3, block=16, seed=1, nchain=841)
r = eng.integrate("mcmc", neval=10**8, niter=4, block=16, seed=1, first_iteration=3, nchain=841)
ms, wg, th = eng.kernel_times_ms(4)
print("%-40s kernel %.2f ms  (%d x %d)" % (os.environ.get("MCI_JIT_FLAGS", "baseline"), float(np.median(ms)), wg, th))
