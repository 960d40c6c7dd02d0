#!/usr/bin/env python3
"""Development tool (GPU box): occupancy of the lane-per-chain kernels -- workgroup size and chains per block forced (more waves per SIMD
need more chains), registers capped through MCI_JIT_FLAGS=-DMCI_CHAIN_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(N,N))).
usage: python tools/chain_occupancy.py c3|c5 threads nchain_per_block"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mcintegration_jl_amd as mci
from mcintegration_jl_amd import isa_mix

name, threads, nchain = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
PI = math.pi
if name == "c3":
    p = mci.catalog.bubble_parameters()
    var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
           mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
    cfg, f, meas, solver, kern = mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)]), mci.catalog.bubble(), mci.bin_by(4), "vegasmc", "mci_vegasmc_chains"
else:
    cfg, f, meas, solver, kern = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss(), None, "mcmc", "mci_mcmc_chains"
eng = mci.Engine(cfg, f, measure=meas, threads=threads if threads else None)
eng.set_chain_speculation(1)
eng.set_kernel_timing(1)
eng.compile(solver)
res = isa_mix.resources(eng.code_object(solver)).get(kern)
eng.integrate(solver, neval=10**8, niter=6, block=16, seed=1, nchain=nchain)
r = eng.integrate(solver, neval=10**8, niter=6, block=16, seed=1, first_iteration=6, ignore=0, nchain=nchain)
ms, wg, th = eng.kernel_times_ms(6)
print("%s threads=%d nchain=%d flags=%r: kernel %.3f ms (wg=%d th=%d) lds %d B  %s" % (name, threads, nchain, os.environ.get("MCI_JIT_FLAGS", ""), float(np.median(ms)), wg, th, eng.lds_bytes, res), flush=True)
mci.shutdown()
