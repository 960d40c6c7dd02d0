#!/usr/bin/env python3
"""Bias of the chain solvers' estimate at ONE configuration, several ways (GPU box; VERDICT r04 "what's weak" 1):

    python tools/bias_ab.py full <case> <solver> [nseeds] [neval] [niter] [block]
        cold integrate(neval, niter) over seeds, automatic chain counts (the product's default call): pooled and unweighted
        deviations, and the deviation ITERATION BY ITERATION (fresh iteration 1 against the carried iterations behind it)
    python tools/bias_ab.py ab <case> <solver> [nseeds] [neval] [niter] [block] [nproc]
        the reference's chain (nchain = 1: one chain per block, the oracle's chain bit for bit up to reassociation,
        tests/test_hip_parity.py) against the automatic many-chain decomposition on the SAME (neval, block, niter, seeds): per arm the
        pooled deviation, the unweighted one, scatter / reported error, and the difference of the two arms' means in units of its
        error.  `nproc` worker processes share the GPU (nchain = 1 runs are a few waves each).
cases: c5 | bubble | cos | x2 | sphere2 | hyper | log  (tools/mcmc_policy.py)"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def run_seeds(job):
    """one worker: cold integrate() for every seed of its share; returns per seed (weighted mean, reported error, iteration means)"""
    name, solver, seeds, neval, niter, block, nchain = job
    import mcintegration_jl_amd as mci
    from mcmc_policy import case
    out = []
    t0 = time.perf_counter()
    carry = os.environ.get("BIAS_CARRY")        # "off": every iteration starts its chains afresh (mci_set_chain_carry(prob, 0))
    lanes = os.environ.get("BIAS_LANES")        # lanes per chain (mci_set_chain_speculation): "1" = one lane per chain
    if os.environ.get("BIAS_FRESH_FLOORS"):     # length of automatic :vegasmc chains started afresh, in burn-in floors (mci_debug_override)
        from mcintegration_jl_amd._lib import lib, check
        check(lib().mci_debug_override(b"fresh_floors", int(os.environ["BIAS_FRESH_FLOORS"]), 1))
    if os.environ.get("BIAS_FRESH_BURNIN"):     # ... and the part of such a chain that is not measured, in per cent
        from mcintegration_jl_amd._lib import lib, check
        check(lib().mci_debug_override(b"fresh_burnin_pct", int(os.environ["BIAS_FRESH_BURNIN"]), 1))
    for seed in seeds:
        cfg, f, meas, exact = case(name, seed=seed)
        eng = mci.Engine(cfg, f, measure=meas)
        if carry:
            eng.set_chain_carry(carry)
        if lanes:
            eng.set_chain_speculation(int(lanes))
        r = eng.integrate(solver, neval=neval, niter=niter, block=block, seed=seed, nchain=nchain)   # (ignore = 1: the library's default with adapt)
        out.append((seed, np.array(r["mean"]), np.array(r["stdev"]), np.array(r["iter_mean"]).reshape(niter, -1), np.array(r["iter_std"]).reshape(niter, -1),
                    int(r["warmup"]), 1))
        eng.close()
    return out, time.perf_counter() - t0


def gather(name, solver, nseeds, neval, niter, block, nchain, nproc):
    seeds = list(range(1, nseeds + 1))
    if nproc <= 1:
        rows, secs = run_seeds((name, solver, seeds, neval, niter, block, nchain))
    else:
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        jobs = [(name, solver, seeds[i::nproc], neval, niter, block, nchain) for i in range(nproc)]
        t0 = time.perf_counter()
        with ctx.Pool(nproc) as pool:
            parts = pool.map(run_seeds, jobs)
        secs = time.perf_counter() - t0
        rows = sorted((r for part, _ in parts for r in part), key=lambda r: r[0])
    return rows, secs


def summarize(rows, exact):
    ms = np.array([r[1] for r in rows])
    es = np.array([r[2] for r in rows])
    ig = rows[0][6]
    im = np.array([r[3] for r in rows])            # [seed][iteration][obs]
    us = im[:, ig:].mean(1)                        # plain mean of the counted iterations
    n = len(rows)
    exact = np.ravel(np.array(exact, dtype=float))[:ms.shape[1]]
    pooled = (ms.mean(0) - exact) / (np.sqrt((es ** 2).sum(0)) / n)
    unw = (us.mean(0) - exact) / (us.std(0, ddof=1) / math.sqrt(n))
    scat = ms.std(0, ddof=1) / np.sqrt((es ** 2).mean(0))
    per_run = (ms.mean(0) - exact) / np.sqrt((es ** 2).mean(0))
    per_iter = (im.mean(0) - exact) / (im.std(0, ddof=1) / math.sqrt(n))   # [iteration][obs]: seed scatter as the error
    return dict(ms=ms, es=es, us=us, pooled=pooled, unw=unw, scat=scat, per_run=per_run, per_iter=per_iter, exact=exact)


def fmt(v):
    return np.array2string(np.round(np.asarray(v), 2), separator=" ")


def main():
    mode = sys.argv[1]
    name, solver = sys.argv[2], sys.argv[3]
    nseeds = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    neval = int(float(sys.argv[5])) if len(sys.argv) > 5 else 10**8
    niter = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    block = int(sys.argv[7]) if len(sys.argv) > 7 else 16
    nproc = int(sys.argv[8]) if len(sys.argv) > 8 else 1
    from mcmc_policy import case
    exact = case(name)[3]
    if mode == "full":
        nchain_full = int(os.environ.get("BIAS_NCHAIN", "0"))   # chains per block (0 = automatic)
        rows, secs = gather(name, solver, nseeds, neval, niter, block, nchain_full, nproc)
        s = summarize(rows, exact)
        print("%s :%s  %d seeds x cold integrate(neval=%.0e, niter=%d, block=%d), %s, carry=%s; %.2f s per run" % (
            name, solver, nseeds, neval, niter, block, "nchain=%d per block" % nchain_full if nchain_full else "automatic chain counts", os.environ.get("BIAS_CARRY", "auto"), secs / nseeds))
        print("  pooled (weighted mean - exact) / pooled reported error : %s" % fmt(s["pooled"]))
        print("  unweighted mean of the counted iterations, seed-scatter error: %s" % fmt(s["unw"]))
        print("  mean deviation per run in units of one run's error     : %s" % fmt(s["per_run"]))
        print("  seed scatter / reported error                          : %s" % fmt(s["scat"]))
        print("  iteration by iteration (mean over seeds - exact) / (seed scatter / sqrt(n)):")
        for i, row in enumerate(s["per_iter"]):
            print("    iteration %2d%s: %s" % (i + 1, " (ignored)" if i < rows[0][6] else "          ", fmt(row)))
    else:
        arms = {}
        for label, nchain in (("reference chain (nchain=1)", 1), ("automatic chains", 0)):
            rows, secs = gather(name, solver, nseeds, neval, niter, block, nchain, nproc)
            arms[label] = summarize(rows, exact)
            s = arms[label]
            print("%s :%s  %s: %d seeds x cold integrate(neval=%.0e, niter=%d, block=%d); %.2f s per run (%d processes)" % (name, solver, label, nseeds, neval, niter, block, secs / nseeds, nproc))
            print("  pooled (weighted mean - exact) / pooled reported error : %s" % fmt(s["pooled"]))
            print("  unweighted mean of the counted iterations, seed-scatter error: %s" % fmt(s["unw"]))
            print("  mean deviation per run in units of one run's error     : %s" % fmt(s["per_run"]))
            print("  seed scatter / reported error                          : %s" % fmt(s["scat"]), flush=True)
        a, b = arms["reference chain (nchain=1)"], arms["automatic chains"]
        n = len(a["ms"])
        d = (b["ms"].mean(0) - a["ms"].mean(0)) / np.sqrt(a["ms"].var(0, ddof=1) / n + b["ms"].var(0, ddof=1) / n)
        du = (b["us"].mean(0) - a["us"].mean(0)) / np.sqrt(a["us"].var(0, ddof=1) / n + b["us"].var(0, ddof=1) / n)
        print("  automatic - reference chain, weighted means, in units of the difference's seed-scatter error : %s" % fmt(d))
        print("  automatic - reference chain, unweighted means                                                : %s" % fmt(du), flush=True)
    import mcintegration_jl_amd as mci
    mci.shutdown()


if __name__ == "__main__":
    main()
