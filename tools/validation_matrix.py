#!/usr/bin/env python3
"""Every catalog integrand with a known answer x the three solvers x several seeds, automatic chain counts:
5 training + 10 production iterations per run; per (integrand, solver) the pooled deviation from the exact value in units
of the pooled error, the largest single-run deviation, and the ratio of the seed scatter to the reported error
(1 = the error bars are honest).   usage: python tools/validation_matrix.py [nseeds] [neval] [names] [solvers] [block] [nchain] [rng rounds]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mcintegration_jl_amd as mci
from catalog_params import bubble_exact_finite_T, genz_exact
PI = math.pi
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NE = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**7
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None     # substrings of integrand names
SOLVERS = sys.argv[4].split(",") if len(sys.argv) > 4 else ["vegas", "vegasmc", "mcmc"]
BLOCK = int(sys.argv[5]) if len(sys.argv) > 5 else 16
NCHAIN = int(sys.argv[6]) if len(sys.argv) > 6 else 0
ROUNDS = int(sys.argv[7]) if len(sys.argv) > 7 else 10     # Philox4x32 rounds (mci_set_rng_rounds): 10 | 7
GAMMA = float(os.environ.get("VAL_GAMMA", "1.0"))           # doReweight! exponent (main.jl:80); 0 freezes the reweight factors
L = math.sqrt(50.0)
p = mci.catalog.bubble_parameters()

def bub():
    var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
           mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
    return mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])

def hyper_exact():
    """analytic: V_d * (d/(2 pi e))^(d/2) sqrt(d) sqrt(pi), d = 2, 3, 4 -> 0.922137, 0.946661, 0.959502.  The constants in the
    reference's test (test/montecarlo.jl:333: 0.9230, 0.94724, 0.96118) are ~1e-3 off, inside its 7 sigma at its neval."""
    out = []
    for d in (2, 3, 4):
        out.append(math.pi ** (d / 2) / math.gamma(d / 2 + 1) * (d / (2 * math.pi * math.e)) ** (d / 2) * math.sqrt(d) * math.sqrt(math.pi))
    return out

CASES = [
    ("x^2+y^2", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), mci.catalog.x2y2(), None, [2.0 / 3.0], 0.0),
    ("log(x)/sqrt(x)", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]), mci.catalog.log_over_sqrt(), None, [-4.0], 0.0),
    ("sphere1", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), mci.catalog.sphere1(), None, [PI / 4], 0.0),
    ("sphere2 (2 integrands)", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]]), mci.catalog.sphere2(), None, [PI / 4, PI / 6], 0.0),
    ("1/(1-cos cos cos)", lambda: mci.Configuration(var=mci.Continuous(0.0, PI), dof=[[3]]), mci.catalog.singular2(), None, [1.3932039296856769], 0.0),
    ("discrete id 1..3", lambda: mci.Configuration(var=mci.Discrete(1, 3), dof=[[1]]), mci.catalog.discrete_id(), None, [6.0], 0.0),
    ("hypersphere (3)", lambda: mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[2], [3], [4]]), mci.catalog.hypersphere(3), None, hyper_exact(), 0.0),
    ("C2 gaussian16", lambda: mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]]), mci.catalog.gaussian(16), None, [math.erf(5.0) ** 16], 0.0),
    ("C4 genz32", lambda: mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]]), mci.catalog.genz_product_peak(32), None, [genz_exact(32)], 0.0),
    ("C5 nested gauss (4)", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]), mci.catalog.nested_gauss(), None,
     [math.erf(5.0) ** d for d in (3, 6, 9, 12)], 0.0),
    ("C3 bubble (4 q)", bub, mci.catalog.bubble(), mci.bin_by(4), bubble_exact_finite_T(), 0.0),
]
print("%d seeds x (5 training + 10 production iterations x %.0e), block=%d, nchain=%d, Philox4x32-%d; deviations in units of the reported error" % (nseeds, NE, BLOCK, NCHAIN, ROUNDS))
print("%-24s %-8s %-28s %-10s %-14s %s" % ("integrand", "solver", "pooled (mean-exact)/err", "max |dev|", "scatter/err", "s per run"))
for name, mk, f, meas, exact, exact_tol in CASES:
    exact = np.array(exact, dtype=float)
    if ONLY and not any(o in name for o in ONLY):
        continue
    for solver in SOLVERS:
        ms, es, us, secs = [], [], [], 0.0
        for seed in range(1, nseeds + 1):
            eng = mci.Engine(mk(), f, measure=meas, **({"rng_rounds": ROUNDS} if ROUNDS != 10 else {}))
            if os.environ.get("VAL_RW"):   # initial reweight factors (with VAL_GAMMA=0: the factors of the whole run)
                eng.set_reweight(np.array([float(v) for v in os.environ["VAL_RW"].split(",")]))
            eng.integrate(solver, neval=NE, niter=5, block=BLOCK, seed=seed, nchain=NCHAIN, gamma=GAMMA)
            r = eng.integrate(solver, neval=NE, niter=10, block=BLOCK, seed=seed, first_iteration=5, ignore=0, nchain=NCHAIN, gamma=GAMMA)
            ms.append(r["mean"]); es.append(r["stdev"]); secs += r["seconds"]; us.append(r["iter_mean"].mean(0))
        ms, es, us = np.array(ms), np.array(es), np.array(us)
        unw = (us.mean(0) - exact) / (us.std(0, ddof=1) / math.sqrt(nseeds))   # plain mean of the iteration means, error from the seed scatter
        perr = np.sqrt((es ** 2).sum(0)) / nseeds
        perr_eff = np.hypot(perr, exact_tol * np.abs(exact))
        pooled = (ms.mean(0) - exact) / perr_eff
        maxdev = np.max(np.abs(ms - exact) / np.hypot(es, exact_tol * np.abs(exact)))
        scat = ms.std(0, ddof=1) / np.sqrt((es ** 2).mean(0))
        print("%-24s %-8s %-28s %-10.2f %-14s %.3f   unweighted: %s" % (name, solver, np.array2string(np.round(pooled, 2), separator=" "), maxdev,
                                                      np.array2string(np.round(scat, 2), separator=" "), secs / nseeds,
                                                      np.array2string(np.round(unw, 2), separator=" ")), flush=True)
