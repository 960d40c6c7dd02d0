import sys, math, numpy as np
sys.path.insert(0,'/root/repo')
import mcintegration_jl_amd as mci
for name, cfgf in (("shared", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[32]])),
                   ("32grids", lambda: mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]]))):
    for th in (256, 512, 1024):
        try:
            eng = mci.Engine(cfgf(), mci.catalog.genz_product_peak(32), threads=th)
            eng.integrate("vegas", neval=10**8, niter=3, block=16, seed=1)
            r = eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=1, first_iteration=3)
            ms, wg, t = eng.kernel_times_ms(5)
            print(name, "threads", th, "mode", eng.table_mode, "lds", eng.lds_bytes, "kernel ms", np.median(ms), "wg", wg, "G/s", 5e8 / r["seconds"] / 1e9, flush=True)
        except Exception as e:
            print(name, th, "failed", e)
