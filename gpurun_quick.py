import sys, time, math
sys.path.insert(0, '.')
import numpy as np
import mcintegration_jl_amd as mci
L = math.sqrt(50.0)
cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1)
eng = mci.Engine(cfg, mci.catalog.gaussian(16))
eng.compile()
for neval in (10**6, 10**7, 10**8):
    r = eng.integrate("vegas", neval=neval, niter=5, block=16, seed=1)
    ms, wg, th = eng.last_kernel_ms()
    print(neval, "sec", r["seconds"], "Msamples/s", neval*5/r["seconds"]/1e6, "kernel ms", ms, wg, th, "mean", r["mean"], r["stdev"], flush=True)
print(r["iter_mean"].ravel(), r["iter_std"].ravel())
