#!/usr/bin/env python3
"""Development tool (GPU box): throughput of the other BASELINE configs (parity-test cases, not bench lines): per configuration the
cold end-to-end rate of one integrate(niter=10) call on a fresh problem and the trained-map rate."""
import math
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mcintegration_jl_amd as mci
from catalog_params import bubble_exact_finite_T, genz_exact

L = math.sqrt(50.0)
PI = math.pi


def run(name, cfg, f, solver, neval, exact, measure=None, niter_train=5, niter=5, nchain=0):
    """Two numbers per configuration: COLD = a fresh problem's integrate(niter=10), the reference's own usage (main.jl:71-90: one call,
    ten iterations from an untrained map; code object from the kernel cache, every launch -- warm-up launches of the chain solvers
    included -- inside the wall time), and TRAINED = `niter` more iterations continuing from the trained map (bench.py's protocol)."""
    import time
    eng = mci.Engine(cfg, f, measure=measure)
    eng.set_kernel_timing(1)   # kernel durations for every launch, the small ones included
    eng.compile(solver)
    t0 = time.perf_counter()
    c = eng.integrate(solver, neval=neval, niter=10, block=16, seed=1, nchain=nchain)
    cold_wall = time.perf_counter() - t0
    cms = eng.kernel_times_ms(16)[0]
    r = eng.integrate(solver, neval=neval, niter=niter, block=16, seed=1, first_iteration=10, ignore=0, nchain=nchain)
    ms, wg, th = eng.kernel_times_ms(niter)
    dev = (r["mean"] - np.atleast_1d(exact)) / r["stdev"]
    cdev = (c["mean"] - np.atleast_1d(exact)) / c["stdev"]
    print("%-28s mode=%d lds=%6d B  COLD integrate(niter=10): %8.1f ms = %8.1f Msamples/s end to end (library %.1f ms, %d warm-up launches; kernels, ms: %s; dev_sigma=%s)" % (
        name, eng.table_mode, eng.lds_bytes, cold_wall * 1e3, neval * 10 / cold_wall / 1e6, c["seconds"] * 1e3, c.get("warmup", 0),
        " ".join("%.2f" % v for v in cms), np.array2string(cdev, precision=2)), flush=True)
    print("%-28s                      TRAINED: %8.1f Msamples/s  kernel %.3f ms (wg=%d,th=%d)  mean=%s +- %s  dev_sigma=%s" % (
        "", neval * niter / r["seconds"] / 1e6, float(np.median(ms)), wg, th,
        np.array2string(r["mean"], precision=8), np.array2string(r["stdev"], formatter={"float_kind": lambda v: "%.2e" % v}), np.array2string(dev, precision=2)), flush=True)
    eng.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c2i", "c4", "c3v", "c3mc", "c5", "cuba", "c1"]
    if "c5" in which:  # BASELINE configs[4]: 4 integrals on a 12-D pool, :mcmc
        ex = [math.erf(5.0) ** d for d in (3, 6, 9, 12)]
        run("C5 nested gauss mcmc 1e8", mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss(),
            "mcmc", 10**8, ex)
        run("C5 nested gauss vegasmc 1e8", mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss(),
            "vegasmc", 10**8, ex)
        run("C5 nested gauss vegas 1e8", mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss(),
            "vegas", 10**8, ex)
    if "cuba" in which:  # the reference's only printed wall time: example/benchmark/cuba/benchmark.jl:119-158 (0.246 s / 0.495 s)
        import time
        for alg in ("vegas", "vegasmc"):
            mci.integrate(mci.catalog.cuba11(), dof=[[3]] * 11, neval=1e4, solver=alg, seed=1)
            t0 = time.perf_counter()
            r = mci.integrate(mci.catalog.cuba11(), dof=[[3]] * 11, neval=1e5, solver=alg, seed=2)
            print("CUBA11 %-8s neval=1e5 x 10: %.4f s wall   Integral 1 = %.6f +- %.6f ... Integral 11 = %.6f +- %.6f" % (
                alg, time.perf_counter() - t0, r.mean[0], r.stdev[0], r.mean[10], r.stdev[10]), flush=True)
    if "c1" in which:
        run("C1 log/sqrt 1e7", mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]), mci.catalog.log_over_sqrt(), "vegas", 10**7, -4.0)
    if "c2" in which:
        run("C2 gauss16 shared", mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]]), mci.catalog.gaussian(16), "vegas", 10**8, 1.0)
    if "c2i" in which:
        run("C2 gauss16 16 grids", mci.Configuration(var=mci.Continuous([(-L, L)] * 16), dof=[[1]]), mci.catalog.gaussian(16), "vegas", 10**8, 1.0)
    if "c4" in which:
        run("C4 genz32 32 grids", mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]]), mci.catalog.genz_product_peak(32), "vegas",
            10**8, genz_exact(32))
    p = mci.catalog.bubble_parameters()
    # what the bubble integrand integrates to at beta*EF = 25 (the reference's lindhard() is the T = 0 closed form, ~1e-4 away)
    BUB_EXACT = bubble_exact_finite_T()

    def bub():
        var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
               mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
        return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])
    if "c3v" in which:
        run("C3 bubble vegas 1e8", bub(), mci.catalog.bubble(), "vegas", 10**8, BUB_EXACT, measure=mci.bin_by(4))
    if "c3mc" in which:
        run("C3 bubble vegasmc 1e8", bub(), mci.catalog.bubble(), "vegasmc", 10**8, BUB_EXACT, measure=mci.bin_by(4))
        run("C3 bubble vegasmc 1e6", bub(), mci.catalog.bubble(), "vegasmc", 10**6, BUB_EXACT, measure=mci.bin_by(4))
