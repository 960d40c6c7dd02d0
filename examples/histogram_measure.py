#!/usr/bin/env python3
"""The reference's "Measure Histogram" example (docs/src/index.md, Example 6) as it is written there, in Python (needs an MI355X):
the circle's and the sphere's section at radius r, r looked up in `config.userdata` by a Discrete draw, one observable bin per r.
Both closures are traced into the kernels: the table goes into the userdata vector (`ud[(int)bin]`), `obs[i][bin] += w` becomes an
add to a bin chosen at run time.  0-based indices: the Discrete draw runs 1 .. N like the reference's, so the bin is `bin[0] - 1`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcintegration_jl_amd as mci

N = 20
grid = [i / N for i in range(1, N + 1)]


def integrand(vars, config):
    grid = config.userdata            # radius
    x, bin = vars                     # unpack the variables
    r = grid[bin[0] - 1]              # binned variable in [0, 1)
    r1 = x[0] ** 2 + r ** 2 < 1       # circle
    r2 = x[0] ** 2 + x[1] ** 2 + r ** 2 < 1   # sphere
    return r1, r2


def measure(vars, obs, weights, config):
    x, bin = vars
    obs[0][bin[0] - 1] += weights[0]  # circle
    obs[1][bin[0] - 1] += weights[1]  # sphere


res = mci.integrate(integrand, measure=measure, var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, N)), dof=[[1, 1], [2, 1]],
                    obs=[np.zeros(N), np.zeros(N)], userdata=grid, neval=1e6, print=-1)
eng = res.config._engine
print("integrand:", type(eng.integrand).__name__, "| measure:", type(eng.measure).__name__)
print("%6s  %22s  %10s  %22s  %10s" % ("r", "circle", "exact", "sphere", "exact"))
for b, r in enumerate(grid):
    print("%6.3f  %10.6f +- %8.6f  %10.6f  %10.6f +- %8.6f  %10.6f" % (r, res.mean[0][b], res.stdev[0][b], np.sqrt(1 - r * r),
                                                                         res.mean[1][b], res.stdev[1][b], np.pi * (1 - r * r) / 4))
