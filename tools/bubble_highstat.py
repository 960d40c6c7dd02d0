"""High-statistics bubble (C3) runs against the finite-temperature polarisation (tests/catalog_params.bubble_exact_finite_T):
at 1e8 samples per iteration the engine resolves the difference between the reference's T = 0 closed form
(example/bubble.jl:24-36) and what the integrand integrates to at beta*EF = 25."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mcintegration_jl_amd as mci
from catalog_params import bubble_exact, bubble_exact_finite_T

ft, t0 = np.array(bubble_exact_finite_T()), np.array(bubble_exact())
print("finite-T", ft, "\nT=0     ", t0, flush=True)
p = mci.catalog.bubble_parameters()
for solver, ne in (("vegas", 10**8), ("vegasmc", 10**8), ("mcmc", 2 * 10**7)):
    allm = []
    for seed in range(1, 7):
        var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, math.pi, alpha=3.0), mci.Continuous(0.0, 2 * math.pi, alpha=3.0),
               mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
        cfg = mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])
        eng = mci.Engine(cfg, mci.catalog.bubble(), measure=mci.bin_by(4))
        eng.compile(solver)
        eng.integrate(solver, neval=ne, niter=5, block=16, seed=seed)
        r = eng.integrate(solver, neval=ne, niter=10, block=16, seed=seed, first_iteration=5, ignore=0)
        m, e = r["mean"], r["stdev"]
        allm.append(m)
        print("%-8s seed %d  dev(finite-T) %s   dev(T=0) %s  chi2 %s  (%.2f s)" % (
            solver, seed, np.round((m - ft) / e, 2), np.round((m - t0) / e, 1), np.round(r["chi2"], 2), r["seconds"]), flush=True)
    allm = np.array(allm)
    print("%-8s mean over seeds - finite-T: %s  (scatter/sqrt(6) %s)" % (solver, allm.mean(0) - ft, allm.std(0, ddof=1) / math.sqrt(6)), flush=True)
