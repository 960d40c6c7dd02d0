"""A closure traced into the kernels against the hand-written body of the same integrand, trained map, ms per iteration (needs an
MI355X): BASELINE configs[1] (16-D Gaussian, :vegas, neval = 1e8) and configs[4] (nested Gaussians, all three solvers)."""
import math
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import mcintegration_jl_amd as mci
from mcintegration_jl_amd import Continuous, integrate


def timed(f, mk, solver, neval, **kw):
    r = integrate(f, var=mk(), solver=solver, neval=neval / 10, niter=5, print=-1, seed=3, **kw)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        r = integrate(f, config=r.config, solver=solver, neval=neval, niter=10, print=-1)
        best = min(best, (time.time() - t0) / 10)
    return best * 1e3, r


def main():
    L = math.sqrt(50.0)
    g16 = lambda x, c: np.exp(-0.5 * np.sum(x * x)) * (2.0 * np.pi) ** -8.0
    for name, f in (("catalog body", mci.catalog.gaussian(16)), ("closure", g16)):
        ms, r = timed(f, lambda: Continuous(-L, L), "vegas", 1e8, dof=[[16]])
        print("C2 16-D Gaussian :vegas   %-13s %7.3f ms per iteration   %.7f +- %.1e" % (name, ms, r.mean[0], r.stdev[0]), flush=True)


    def nested(x, c):                                   # BASELINE configs[4] as the catalog writes it: one factor per dimension
        out, p = [], 1.0
        for d in range(12):
            p = p * (np.exp(-100.0 * (x[d] - 0.5) * (x[d] - 0.5)) * np.sqrt(100.0 / np.pi))
            if d % 3 == 2:
                out.append(p)
        return tuple(out)


    for solver in ("vegas", "vegasmc", "mcmc"):
        for name, f in (("catalog body", mci.catalog.nested_gauss()), ("closure", (lambda idx, x, c: nested(x, c)[idx]) if solver == "mcmc" else nested)):
            ms, r = timed(f, lambda: Continuous(0.0, 1.0), solver, 1e8, dof=[[3], [6], [9], [12]])
            print("C5 nested Gaussians :%-8s %-13s %7.3f ms per iteration   %s" % (solver, name, ms, np.round(r.mean, 6)), flush=True)


if __name__ == "__main__":
    main()
