"""BASELINE.json's configurations at their FULL sizes (1e8 samples per iteration on one GPU), checked through
properties that do not need an oracle run of that size:
    * bookkeeping: the neval / normalization columns count every sample exactly;
    * additivity: an iteration over blocks [0, B) is the sum of the iterations over [0, B/2) and [B/2, B)
      (the Philox index is global, blocks are independent; only the order of the sums differs);
    * launch-geometry independence: other workgroup decompositions give the same sums;
    * linearity: scaling the integrand by c scales the observable sums by c and the map-training histogram by c^2;
    * the trained estimate sits within 5 sigma of the analytic value.
Tolerances: 1e-9 relative on sums of 1e8 fp64 terms (reassociation), bit-exact on counters.
"""
import math

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from catalog_params import bubble_exact_finite_T, genz_exact

pytestmark = pytest.mark.gpu
PI = math.pi
SEED = 20240229
NEVAL = 10**8
BLOCK = 16
NPB = NEVAL // BLOCK


def split(packed, eng, ni):
    """(statistics head, histogram section) of [stats | histograms | propose | accept]"""
    nstat = 2 * eng.nobs + 2 + ni + 1
    npa = 3 * (ni + 1) * max(ni + 1, len(eng.config.var))
    return packed[:nstat], packed[nstat:len(packed) - 2 * npa]


def check_additive(eng, solver, ni, nchain=0, rtol=1e-9):
    whole = eng.iteration(solver, NPB, 0, BLOCK, iteration=3, seed=SEED, nchain=nchain)
    lo = eng.iteration(solver, NPB, 0, BLOCK // 2, iteration=3, seed=SEED, nchain=nchain)
    hi = eng.iteration(solver, NPB, BLOCK // 2, BLOCK, iteration=3, seed=SEED, nchain=nchain)
    n = eng.nobs
    ws, wh = split(whole, eng, ni)
    ls, lh = split(lo, eng, ni)
    hs, hh = split(hi, eng, ni)
    np.testing.assert_allclose(ls[:2 * n] + hs[:2 * n], ws[:2 * n], rtol=rtol, atol=1e-300)   # sums of block means and of their squares
    # counters: each part carries the clearStatistics offsets of its own (blocks + 1) configs
    np.testing.assert_allclose(ls[2 * n:] + hs[2 * n:], ws[2 * n:], rtol=1e-12, atol=1e-6)
    np.testing.assert_allclose(lh + hh, wh, rtol=rtol, atol=1e-7 * float(np.max(wh)))
    return ws, wh


def test_c2_gaussian16_full_size_properties():
    L = math.sqrt(50.0)
    cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.gaussian(16))
    eng.integrate("vegas", neval=NEVAL, niter=5, block=BLOCK, seed=SEED)            # train the grid
    ws, wh = check_additive(eng, "vegas", 1)
    n = eng.nobs
    assert ws[2 * n + 1] == NEVAL                                                   # config.neval, exact
    assert abs(ws[2 * n] - NEVAL) < 1e-6                                            # normalization = count + 1e-10 offsets
    # other launch geometries, same sums
    for threads, wpb in ((512, 16), (64, 100)):
        e2 = mci.Engine(mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=SEED), mci.catalog.gaussian(16),
                        threads=threads, wg_per_block=wpb)
        e2.set_grid(0, eng.grid(0))
        o = e2.iteration("vegas", NPB, 0, BLOCK, iteration=3, seed=SEED)
        os_, oh = split(o, e2, 1)
        np.testing.assert_allclose(os_, ws, rtol=1e-9)
        np.testing.assert_allclose(oh, wh, rtol=1e-9)
    # linearity in the integrand
    c = 3.0
    e3 = mci.Engine(mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=SEED),
                    mci.Integrand("double s = 0.0; for (int i = 0; i < 16; ++i) s += x[i] * x[i];\n"
                                  "w[0] = 3.0 * exp(-0.5 * s) * pow(2.0 * M_PI, -8.0);"))
    e3.set_grid(0, eng.grid(0))
    o = e3.iteration("vegas", NPB, 0, BLOCK, iteration=3, seed=SEED)
    os_, oh = split(o, e3, 1)
    np.testing.assert_allclose(os_[0], c * ws[0], rtol=1e-9)
    np.testing.assert_allclose(os_[1], c * c * ws[1], rtol=1e-9)
    off = (BLOCK + 1) * 1e-10
    np.testing.assert_allclose(oh - off, c * c * (wh - off), rtol=1e-8, atol=1e-9 * float(np.max(wh)))
    # and the answer
    r = eng.integrate("vegas", neval=NEVAL, niter=5, block=BLOCK, seed=SEED, first_iteration=5, ignore=0)
    exact = math.erf(L / math.sqrt(2.0)) ** 16
    assert abs(r["mean"][0] - exact) < 5 * r["stdev"][0] and r["stdev"][0] < 2e-5, (r["mean"], r["stdev"])


def test_c4_genz32_full_size_properties():
    cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.genz_product_peak(32))
    assert eng.table_mode == 3                                                      # 32 grids: L2 gathers, 2 histogram tiles, split pass
    eng.integrate("vegas", neval=NEVAL, niter=5, block=BLOCK, seed=SEED)
    ws, wh = check_additive(eng, "vegas", 1)
    n = eng.nobs
    assert ws[2 * n + 1] == NEVAL
    # every one of the 32 grids' histograms received every sample's weight: equal totals (offsets removed)
    per_grid = wh.reshape(32, -1).sum(axis=1) - (BLOCK + 1) * 1e-10 * wh.size / 32
    np.testing.assert_allclose(per_grid, per_grid[0], rtol=1e-9)
    r = eng.integrate("vegas", neval=NEVAL, niter=5, block=BLOCK, seed=SEED, first_iteration=5, ignore=0)
    assert abs(r["mean"][0] - genz_exact(32)) < 5 * r["stdev"][0], (r["mean"], r["stdev"], genz_exact(32))
    # the launch in chunks (a bounded parked stream, vegas/montecarlo.jl:117-187: constant memory in neval) is the one-chunk launch up to
    # the order of its sums: same samples, same Philox indices, partial rows added chunk by chunk
    from mcintegration_jl_amd._lib import check, lib
    one = eng.iteration("vegas", NPB, 0, BLOCK, iteration=11, seed=SEED)
    assert eng.split_chunks() == (1, NEVAL * 48)                                    # 8 B of weight + 32 bins of 10 bits per parked sample
    check(lib().mci_debug_override(b"split_chunk", 2 ** 25, 1))
    try:
        three = eng.iteration("vegas", NPB, 0, BLOCK, iteration=11, seed=SEED)
        assert eng.split_chunks()[0] == 3
    finally:
        check(lib().mci_debug_override(b"split_chunk", 0, 0))
    np.testing.assert_allclose(three[:4], one[:4], rtol=1e-12)
    np.testing.assert_allclose(three[4:], one[4:], rtol=1e-10)


def test_c3_bubble_vegasmc_full_size_properties():
    p = mci.catalog.bubble_parameters()
    var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
           mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
    cfg = mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.bubble(), measure=mci.bin_by(4))
    eng.integrate("vegasmc", neval=NEVAL, niter=5, block=BLOCK, seed=SEED)
    nchain = 2441                                                                   # what the automatic setting picks at this size
    ws, wh = check_additive(eng, "vegasmc", 1, nchain=nchain, rtol=1e-8)
    n = eng.nobs
    steps = NPB // nchain
    assert abs(ws[2 * n + 1] - BLOCK * nchain * steps) <= 0.01 * NEVAL              # config.neval counts the proposals that were evaluated
    vis = ws[2 * n + 2:2 * n + 4]
    assert np.all(vis > 0)
    r = eng.integrate("vegasmc", neval=NEVAL, niter=5, block=BLOCK, seed=SEED, first_iteration=5, ignore=0)
    ft = np.array(bubble_exact_finite_T())
    assert np.all(np.abs(r["mean"] - ft) < 5 * r["stdev"]), (r["mean"], r["stdev"], ft)


def test_c5_nested_gauss_full_size_properties(oracle):
    """BASELINE configs[4]: 4 integrals sharing a 12-D Continuous pool (dof [[3],[6],[9],[12]]) at 1e8 steps per iteration, under
    all three solvers (the config names :mcmc; :vegasmc / :vegas run the same padding path, variable.jl:628-641).
    :mcmc with the automatic (measured) chain length -- exact counters, additivity over block ranges, the hold histogram of a
    reduced-size launch equal to the oracle's bucket by bucket, products of erf within 5 sigma."""
    exact = np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])

    def engine():
        return mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=SEED), mci.catalog.nested_gauss())

    # ---- :vegas ----
    eng = engine()
    eng.integrate("vegas", neval=NEVAL, niter=5, block=BLOCK, seed=SEED)
    ws, wh = check_additive(eng, "vegas", 4)
    n = eng.nobs
    assert n == 4 and ws[2 * n + 1] == NEVAL and abs(ws[2 * n] - NEVAL) < 1e-6
    r = eng.integrate("vegas", neval=NEVAL, niter=5, block=BLOCK, seed=SEED, first_iteration=5, ignore=0)
    assert np.all(np.abs(r["mean"] - exact) < 5 * r["stdev"]) and np.all(r["stdev"] < 5e-3), (r["mean"], r["stdev"])

    # ---- :vegasmc (automatic chain count) ----
    eng = engine()
    eng.integrate("vegasmc", neval=NEVAL, niter=5, block=BLOCK, seed=SEED)
    nchain = 1017                                          # the automatic choice at this size: neval/block / (8 burn-in floors of 64 * 12 steps)
    ws, wh = check_additive(eng, "vegasmc", 4, nchain=nchain, rtol=1e-8)
    steps = NPB // nchain
    assert abs(ws[2 * n + 1] - BLOCK * nchain * steps) <= 0.01 * NEVAL
    assert np.all(ws[2 * n + 2:2 * n + 2 + 5] > 0)         # visited: every integrand and the normalisation
    r = eng.integrate("vegasmc", neval=NEVAL, niter=5, block=BLOCK, seed=SEED, first_iteration=5, ignore=0)
    assert np.all(np.abs(r["mean"] - exact) < 5 * r["stdev"]), (r["mean"], r["stdev"])

    # ---- :mcmc (the solver BASELINE names; automatic, measured chain length) ----
    eng = engine()
    eng.integrate("mcmc", neval=NEVAL, niter=5, block=BLOCK, seed=SEED)
    nchain = 1024
    ws, wh = check_additive(eng, "mcmc", 4, nchain=nchain, rtol=1e-8)
    steps = NPB // nchain
    # visited counts every step of every chain (mcmc/montecarlo.jl:136), burn-in included: exact integer bookkeeping
    nburn = max(int(math.floor(steps * 0.1)), 64 * 12 + 16 * 2 * 5)
    vis = ws[2 * n + 2:2 * n + 2 + 5] - (BLOCK + 1) * 1e-8
    assert abs(vis.sum() - BLOCK * nchain * (steps + nburn)) < 1e-3 * BLOCK * nchain
    hh = eng.hold_histogram()
    assert hh.sum() == BLOCK // 2 * nchain                 # (the last launch of check_additive ran blocks [8, 16))
    r = eng.integrate("mcmc", neval=NEVAL, niter=5, block=BLOCK, seed=SEED, first_iteration=5, ignore=0)
    assert np.all(np.abs(r["mean"] - exact) < 5 * r["stdev"]), (r["mean"], r["stdev"])
    # hold histogram of a launch the oracle can follow: bucket by bucket
    small = engine()
    ocfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[3], [6], [9], [12]])
    ocfg.set_thermal_ratio(0.1)
    got = small.iteration("mcmc", 40000, 0, 4, iteration=0, seed=SEED, nchain=32, thermal_ratio=0.1)
    ref = ocfg.iteration(oracle.MCMC, "nested_gauss", [4.0, 3.0, 6.0, 9.0, 12.0], 40000, 0, 4, 0, SEED, nchain=32)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    np.testing.assert_array_equal(small.hold_histogram(), ocfg.hold_hist)
