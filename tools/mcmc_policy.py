#!/usr/bin/env python3
"""Development tool (GPU box): the automatic chain-length policy of solver = :mcmc and the error bars of carried chains.

    python tools/mcmc_policy.py trace <case> [neval] [niter]      per-iteration table of a COLD problem: chains per block, chain length,
                                                                  kernel ms, top bucket of the holding-time histogram the launch measured
    python tools/mcmc_policy.py cold <case> [neval] [niter] [reps] wall time of a fresh integrate() call (code object from the cache), every launch included
    python tools/mcmc_policy.py stats <cases> [nseeds] [neval] [niter] [solver]
                                                                  cold calls over seeds: pooled pull, scatter / reported error (block-lineage error
                                                                  for carried chains) and scatter / the reference's statistics.jl:198 error
cases: c5 | bubble | cos | x2 | sphere2 | hyper | log   (comma-separated for stats)"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mcintegration_jl_amd as mci
from catalog_params import bubble_exact_finite_T

PI = math.pi


def case(name, seed=1):
    if name == "c5":
        return (mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=seed), mci.catalog.nested_gauss(), None,
                [math.erf(5.0) ** d for d in (3, 6, 9, 12)])
    if name == "bubble":
        p = mci.catalog.bubble_parameters()
        var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
               mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
        return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=seed), mci.catalog.bubble(), mci.bin_by(4), bubble_exact_finite_T()
    if name == "cos":
        return mci.Configuration(var=mci.Continuous(0.0, PI), dof=[[3]], seed=seed), mci.catalog.singular2(), None, [1.3932039296856769]
    if name == "x2":
        return mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]], seed=seed), mci.catalog.x2y2(), None, [2.0 / 3.0]
    if name == "sphere2":
        return mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], seed=seed), mci.catalog.sphere2(), None, [PI / 4, PI / 6]
    if name == "log":
        return mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]], seed=seed), mci.catalog.log_over_sqrt(), None, [-4.0]
    if name == "hyper":
        ex = [math.pi ** (d / 2) / math.gamma(d / 2 + 1) * (d / (2 * math.pi * math.e)) ** (d / 2) * math.sqrt(d) * math.sqrt(math.pi) for d in (2, 3, 4)]
        return mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[2], [3], [4]], seed=seed), mci.catalog.hypersphere(3), None, ex
    raise SystemExit("unknown case %r" % name)


def trace(name, neval, niter, solver="mcmc"):
    cfg, f, meas, exact = case(name)
    eng = mci.Engine(cfg, f, measure=meas)
    if os.environ.get("POLICY_LANES"):   # lanes per chain (mci_set_chain_speculation): 1 = one lane per chain always
        eng.set_chain_speculation(int(os.environ["POLICY_LANES"]))
    else:
        eng.compile(solver)
    eng.set_kernel_timing(1)
    block = 16
    npb = neval // block
    print("%s %s neval=%.0e block=%d, cold problem; per iteration: chains/block, steps/chain, carried, kernel ms, top hold bucket (2^b)" % (name, solver, neval, block))
    tot = 0.0
    for it in range(niter):
        eng.run(mci._lib.SOLVERS[solver], npb, 0, block, it, 1, 1, 0, 0.1)
        nchain, carried = eng.last_chain_launch()
        hh = eng.hold_histogram() if solver == "mcmc" else np.zeros(64)
        top = int(np.max(np.nonzero(hh)[0])) if hh.any() else -1
        m, e = eng.finish(mci._lib.SOLVERS[solver], block, True, 1.0)
        ms = eng.kernel_times_ms(1)[0]
        tot += float(ms[-1])
        valid, warm = eng.mcmc_launch_valid()[:2] if solver == "mcmc" else (True, True)
        print("  it %2d  nchain %6d  len %8d  carried %d  lanes %2d  kernel %9.3f ms  hold top 2^%d  %s  mean[0] %.6f +- %.1e" % (
            it, nchain, npb // max(nchain, 1), carried, eng.last_chain_speculation()[0], ms[-1], top, "valid" if valid else "short" + ("" if warm else " (warm-up: integrate() would run it again)"),
            np.ravel(m)[0], np.ravel(e)[0]), flush=True)
    print("  sum of kernel times %.1f ms -> %.2f Gsteps/s" % (tot, neval * niter / tot / 1e6))
    eng.close()


def cold(name, neval, niter, reps=3, solver="mcmc"):
    for r in range(reps):
        cfg, f, meas, exact = case(name, seed=r + 1)
        t0 = time.perf_counter()
        res = mci.integrate(f, config=cfg, measure=meas, solver=solver, neval=neval, niter=niter)
        dt = time.perf_counter() - t0
        dev = (np.ravel(res.mean[0]) - np.ravel(exact)[:len(np.ravel(res.mean[0]))]) / np.ravel(res.stdev[0])
        print("%s %s cold integrate(neval=%.0e, niter=%d): %.1f ms wall (library %.1f ms) -> %.2f Gsteps/s end to end; correlated=%s warm-up launches=%d  dev[0]=%s sigma" % (
            name, solver, neval, niter, dt * 1e3, res.seconds * 1e3, neval * niter / dt / 1e9, res.correlated, res.warmup, np.round(dev, 2)), flush=True)
        cfg._engine.close()


def stats(names, nseeds, neval, niter, solver="mcmc", block=16):
    print("%d seeds x cold integrate(solver=%s, neval=%.0e, niter=%d, ignore=1), block=%d, automatic chain counts" % (nseeds, solver, neval, niter, block))
    print("%-8s %-30s %-9s %-26s %-26s %s" % ("case", "pooled (mean-exact)/err", "max|dev|", "scatter/err (reported)", "scatter/err (statistics.jl)", "s/run  warm-up launches/run"))
    for name in names:
        ms, es, er, us, secs, wu = [], [], [], [], 0.0, 0
        for seed in range(1, nseeds + 1):
            cfg, f, meas, exact = case(name, seed=seed)
            t0 = time.perf_counter()
            res = mci.integrate(f, config=cfg, measure=meas, solver=solver, neval=neval, niter=niter, block=block)
            secs += time.perf_counter() - t0
            wu += res.warmup
            ms.append(res._flat_mean)
            es.append(res._flat_std)
            er.append(mci.Result(res.iter_mean, res.iter_std, cfg, res.ignore)._flat_std)
            us.append(res.iter_mean[res.ignore:].mean(0))   # plain mean of the counted iterations: free of the correlation between an iteration's mean and its weight
            cfg._engine.close()
        ms, es, er, us = np.array(ms), np.array(es), np.array(er), np.array(us)
        exact = np.ravel(np.array(exact, dtype=float))[:ms.shape[1]]
        pooled = (ms.mean(0) - exact) / (np.sqrt((es ** 2).sum(0)) / nseeds)
        maxdev = np.max(np.abs(ms - exact) / es)
        scat = ms.std(0, ddof=1) / np.sqrt((es ** 2).mean(0))
        scat_ref = ms.std(0, ddof=1) / np.sqrt((er ** 2).mean(0))
        unw = (us.mean(0) - exact) / (us.std(0, ddof=1) / math.sqrt(nseeds))
        print("%-8s %-30s %-9.2f %-26s %-26s %.3f  %.2f   unweighted pooled: %s" % (name, np.round(pooled, 2), maxdev, np.round(scat, 2), np.round(scat_ref, 2), secs / nseeds, wu / nseeds,
                                                                          np.round(unw, 2)), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "trace":
        trace(sys.argv[2], int(float(sys.argv[3])) if len(sys.argv) > 3 else 10**8, int(sys.argv[4]) if len(sys.argv) > 4 else 10,
              sys.argv[5] if len(sys.argv) > 5 else "mcmc")
    elif mode == "cold":
        cold(sys.argv[2], int(float(sys.argv[3])) if len(sys.argv) > 3 else 10**8, int(sys.argv[4]) if len(sys.argv) > 4 else 10,
             int(sys.argv[5]) if len(sys.argv) > 5 else 3, sys.argv[6] if len(sys.argv) > 6 else "mcmc")
    else:
        stats(sys.argv[2].split(","), int(sys.argv[3]) if len(sys.argv) > 3 else 32, int(float(sys.argv[4])) if len(sys.argv) > 4 else 10**7,
              int(sys.argv[5]) if len(sys.argv) > 5 else 10, sys.argv[6] if len(sys.argv) > 6 else "mcmc", int(sys.argv[7]) if len(sys.argv) > 7 else 16)
    mci.shutdown()
