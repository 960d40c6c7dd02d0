"""Randomised layouts: pools of Continuous / Discrete / CompositeVar with ragged dof tables, 1-3 integrands, grids of
different sizes -- the sample-batch kernels are JIT-specialised per layout, so layout-specific code paths (static draw
tables, slot selects, neighbor graph, table placement) are exercised against the oracle on identical Philox streams."""
import math

import numpy as np
import pytest

import mcintegration_jl_amd as mci

pytestmark = pytest.mark.gpu
SEED = 424242


def random_case(rng):
    npool = int(rng.integers(1, 4))
    ni = int(rng.integers(1, 4))
    var, oleaves, pool_nleaf = [], [], []
    for v in range(npool):
        kind = rng.choice(["cont", "disc", "comp"])
        if kind == "cont":
            lo = float(rng.uniform(-2, 0))
            hi = float(rng.uniform(0.5, 3))
            ninc = int(rng.choice([17, 100, 1000]))
            alpha = float(rng.choice([1.0, 2.0, 3.0]))
            var.append(mci.Continuous(lo, hi, alpha=alpha, ninc=ninc))
            oleaves.append(dict(kind=0, pool=v, lower=lo, upper=hi, npts=ninc, alpha=alpha))
            pool_nleaf.append(1)
        elif kind == "disc":
            lo = int(rng.integers(0, 3))
            hi = lo + int(rng.integers(0, 6))          # K = 1 .. 6 (K = 1: nothing to sample)
            adapt = bool(rng.integers(0, 2))
            var.append(mci.Discrete(lo, hi, adapt=adapt))
            oleaves.append(dict(kind=1, pool=v, lower=lo, upper=hi, adapt=adapt))
            pool_nleaf.append(1)
        else:
            a = (float(rng.uniform(-1, 0)), float(rng.uniform(0.5, 2)))
            b = (int(rng.integers(1, 3)), int(rng.integers(3, 6)))
            var.append(mci.CompositeVar(mci.Continuous(*a), mci.Discrete(*b)))
            oleaves.append(dict(kind=0, pool=v, lower=a[0], upper=a[1]))
            oleaves.append(dict(kind=1, pool=v, lower=b[0], upper=b[1]))
            pool_nleaf.append(2)
    dof = [[int(rng.integers(0, 4)) for _ in range(npool)] for _ in range(ni)]
    for i in range(ni):                                  # every integrand owns at least one draw
        if sum(dof[i]) == 0:
            dof[i][int(rng.integers(0, npool))] = 1
    maxdof = [max(d[v] for d in dof) for v in range(npool)]
    # flat draws (pool, slot, leaf) and which integrand may read which
    draws = [(v, s, l) for v in range(npool) for s in range(maxdof[v]) for l in range(pool_nleaf[v])]
    lines = []
    for i in range(ni):
        own = [k for k, (v, s, l) in enumerate(draws) if s < dof[i][v]]
        coef = rng.uniform(0.2, 1.5, size=len(own))
        arg = " + ".join("%.6f * x[%d]" % (c, k) for c, k in zip(coef, own))
        lines.append("w[%d] = %.3f + 0.5 * cos(%s) + 0.05 * x[%d] * x[%d];" % (i, 1.0 + 0.3 * i, arg, own[0], own[-1]))
    return tuple(var), oleaves, dof, "\n".join(lines), len(draws)


@pytest.mark.parametrize("case_id", range(16))
def test_random_layout_matches_oracle(oracle, case_id):
    rng = np.random.default_rng(1000 + case_id)
    var, oleaves, dof, body, ndraw = random_case(rng)
    cfg = mci.Configuration(var=var, dof=dof, seed=SEED)
    eng = mci.Engine(cfg, mci.Integrand(body))
    assert eng.ndraw == ndraw
    fn = oracle.compile_c_integrand(body)
    for solver, osolver in (("vegas", oracle.VEGAS), ("vegasmc", oracle.VEGASMC), ("mcmc", oracle.MCMC)):
        ocfg = oracle.Config(oleaves, dof)
        got = eng.iteration(solver, 2400, 0, 3, iteration=case_id, seed=SEED, nchain=8)
        ref = ocfg.iteration(osolver, fn, None, 2400, 0, 3, case_id, SEED, nchain=8)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300, err_msg="%s dof=%s\n%s" % (solver, dof, body))
        if solver == "mcmc":   # the holding-time bookkeeping of the automatic chain length, bucket by bucket
            np.testing.assert_array_equal(eng.hold_histogram(), ocfg.hold_hist, err_msg="dof=%s" % dof)
    # and two full iterations with training in between (vegas)
    ocfg = oracle.Config(oleaves, dof)
    r = eng.integrate("vegas", neval=24000, niter=3, block=8, seed=SEED)
    o = ocfg.integrate(oracle.VEGAS, fn, None, neval=24000, niter=3, block=8, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-6)
