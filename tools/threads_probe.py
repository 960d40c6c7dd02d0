"""Workgroup size sweep for the vegas sample pass on C3 (bubble), C5 (nested Gaussians) and C2: does more threads per workgroup (more waves
per SIMD at the same LDS footprint) pay?"""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcintegration_jl_amd as mci
PI = math.pi
p = mci.catalog.bubble_parameters()
L = math.sqrt(50.0)

def bub():
    var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
           mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
    return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])

cases = [("C3 bubble", bub, mci.catalog.bubble(), mci.bin_by(4)),
         ("C5 nested", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss(), None),
         ("C2 gauss16", lambda: mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]]), mci.catalog.gaussian(16), None),
         ("C2i 16 grids", lambda: mci.Configuration(var=mci.Continuous([(-L, L)] * 16), dof=[[1]]), mci.catalog.gaussian(16), None)]
for name, mk, f, meas in cases:
    for th in (0, 128, 256, 512, 1024):
        try:
            eng = mci.Engine(mk(), f, measure=meas, threads=th) if th else mci.Engine(mk(), f, measure=meas)
            eng.integrate("vegas", neval=10**8, niter=3, block=16, seed=1)
            r = eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=1, first_iteration=3)
            ms, wg, t = eng.kernel_times_ms(5)
            print("%-14s threads %-5s (launched %4d x %d)  lds %6d  kernel %.3f ms  %.1f Gsamples/s" % (name, th or "auto", wg, t, eng.lds_bytes, float(np.median(ms)), 5e8 / r["seconds"] / 1e9), flush=True)
        except Exception as e:
            print(name, th, "failed:", str(e)[:80], flush=True)
