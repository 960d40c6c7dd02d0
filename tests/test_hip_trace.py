"""Python closures traced into device source (mcintegration_jl_amd/trace.py) on the GPU: same estimate as the closure on the host callback
path (same samples: the written-out body computes what the closure computes), all three solvers.  The CPU side of the tracer --
bodies against closures through gcc, refusals, the offline gfx950 compile -- is tests/test_trace.py."""
import math

import numpy as np
import pytest

import mcintegration_jl_amd as mci

pytestmark = pytest.mark.gpu


def test_traced_closure_gives_the_host_closure_estimate():
    """vegas/montecarlo.jl:140-144: `integrand(var, config)`; traced = inside the kernel, untraced = batch callback over PCIe"""
    f = lambda x, c: np.exp(-np.sum(x * x, axis=0) / 2) / (2 * np.pi) ** (len(x) / 2)
    exact = math.erf(5.0 / math.sqrt(2.0)) ** 4
    out = []
    for trace in (True, False):
        r = mci.integrate(f, var=mci.Continuous(-5.0, 5.0), dof=[[4]], solver="vegas", neval=200000, niter=8, seed=7, trace=trace, print=-1)
        assert abs(r.mean[0] - exact) < 5 * r.stdev[0] and r.stdev[0] < 2e-4
        out.append(r)
    assert abs(out[0].mean[0] - out[1].mean[0]) < 0.1 * out[0].stdev[0]      # (libm vs the device's exp: rounding level, amplified by train!)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_traced_two_integrand_closure_with_a_select(solver):
    g = lambda x, c: (x[0] ** 2 + x[1] ** 2, mci.trace.where(x[0] > 0.5, x[1], 0.0))
    r = mci.integrate(g, var=mci.Continuous(0.0, 1.0), dof=[[2], [2]], solver=solver, neval=200000, niter=10, seed=7, trace=True, print=-1)
    assert abs(r.mean[0] - 2.0 / 3.0) < 5 * r.stdev[0] and abs(r.mean[1] - 0.25) < 5 * r.stdev[1]
    assert r.stdev[0] < 0.02 and r.stdev[1] < 0.02
