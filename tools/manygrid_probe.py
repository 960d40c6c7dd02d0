"""D independent grids (Genz product peak), D = 8 .. 40: table mode, tiles, throughput and the estimate against the exact value.
(D >= 44: the weight spans ~10^D and the alpha = 2 refinement collapses onto single samples even at 1e8 samples per iteration -- in the
CPU oracle exactly as here; that is the algorithm, not the engine.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mcintegration_jl_amd as mci
from catalog_params import genz_exact
for D in (8, 12, 16, 20, 24, 32, 40):
    cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * D), dof=[[1]], seed=1)
    eng = mci.Engine(cfg, mci.catalog.genz_product_peak(D))
    eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=1)
    r = eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=1, first_iteration=5, ignore=0)
    ms, wg, th = eng.kernel_times_ms(5)
    ex = genz_exact(D)
    print("D=%2d mode=%d lds=%6d  %dx%d  kernel %.3f ms  %.2f Gsamples/s  (mean-exact)/sigma = %+.2f  rel sigma %.1e" % (
        D, eng.table_mode, eng.lds_bytes, wg, th, float(np.median(ms)), 5e8 / r["seconds"] / 1e9, (r["mean"][0] - ex) / r["stdev"][0], r["stdev"][0] / ex), flush=True)
