#!/usr/bin/env python3
"""One BASELINE workload for a profiler to wrap (profiles/collect.sh), or a quick probe of a launch plan:

    python tools/workload.py <c2|c2i|c3|c4|c5> [--solver S] [--neval N] [--niter K] [--nchain C] [--threads T] [--deterministic]

c2: 16-D Gaussian on a shared grid | c2i: on 16 independent grids | c3: example/bubble.jl | c4: 32-D Genz product peak on 32 grids |
c5: 4 nested Gaussians on a 12-D pool.  Default solver: the one BASELINE.json names for the configuration.  Prints the kernel's
HIP-event duration, its launch geometry and the estimate."""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci

L, PI = math.sqrt(50.0), math.pi


def build(name):
    if name == "c2":
        return mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1), mci.catalog.gaussian(16), None, "vegas"
    if name == "c2i":
        return mci.Configuration(var=mci.Continuous([(-L, L)] * 16), dof=[[1]], seed=1), mci.catalog.gaussian(16), None, "vegas"
    if name == "c3":
        p = mci.catalog.bubble_parameters()
        var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
               mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
        return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=1), mci.catalog.bubble(), mci.bin_by(4), "vegasmc"
    if name == "c4":
        return mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=1), mci.catalog.genz_product_peak(32), None, "vegas"
    if name == "c5":
        return mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=1), mci.catalog.nested_gauss(), None, "mcmc"
    raise SystemExit("unknown workload %r" % name)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--solver")
    ap.add_argument("--neval", type=float, default=1e8)
    ap.add_argument("--niter", type=int, default=6)
    ap.add_argument("--nchain", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--cold", action="store_true", help="ONE integrate(niter) call on the fresh problem (the reference's usage, main.jl:71-90) "
                    "and its end-to-end rate, every launch included")
    a = ap.parse_args()
    cfg, f, meas, solver = build(a.workload)
    solver = a.solver or solver
    eng = mci.Engine(cfg, f, measure=meas, deterministic=a.deterministic, **(dict(threads=a.threads) if a.threads else {}))
    eng.set_kernel_timing(1)
    if a.cold:
        import time
        eng.compile(solver)
        t0 = time.perf_counter()
        r = eng.integrate(solver, neval=a.neval, niter=a.niter, block=16, seed=1, nchain=a.nchain)
        wall = time.perf_counter() - t0
        ms, wg, th = eng.kernel_times_ms(a.niter + 8)
        print("%s %s COLD integrate(neval=%.0e, niter=%d): %.1f ms wall, library %.1f ms -> %.2f G/s end to end; %d warm-up launches; sample kernels, ms, launch order: %s" % (
            a.workload, solver, a.neval, a.niter, wall * 1e3, r["seconds"] * 1e3, a.neval * a.niter / wall / 1e9, r.get("warmup", 0), " ".join("%.2f" % v for v in ms)))
        print("   mean=%s +- %s" % (np.array2string(r["mean"], precision=8), np.array2string(r["stdev"], formatter={"float_kind": lambda v: "%.2e" % v})))
        mci.shutdown()
        sys.exit(0)
    half = max(a.niter // 2, 1)
    eng.integrate(solver, neval=a.neval, niter=half, block=16, seed=1, nchain=a.nchain)
    r = eng.integrate(solver, neval=a.neval, niter=a.niter - half or 1, block=16, seed=1, nchain=a.nchain, first_iteration=half, ignore=0)
    ms, wg, th = eng.kernel_times_ms(a.niter)
    extra = "" if solver == "vegas" else "  chains per block %d (carried: %s)" % eng.last_chain_launch()
    print("%s %s neval=%.0e: %.3f s per iteration, sample kernel(s) %.3f ms (median of %d), wg=%d th=%d%s  mean=%s +- %s" % (
        a.workload, solver, a.neval, r["seconds"] / max(a.niter - half, 1), float(np.median(ms)) if len(ms) else float("nan"), len(ms), wg, th, extra,
        np.array2string(r["mean"], precision=8), np.array2string(r["stdev"], precision=2)))
    mci.shutdown()
