"""The reference's own statistical battery (test/montecarlo.jl:298-387, test/bubble.jl:93-133,
test/interface_tests.jl) run through the product's `integrate` on the GPU: |mean - exact| < 7 sigma
(test/runtests.jl:4-9) and the sigma regression bounds.  Reads like the reference's tests."""
import math

import os

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd import Continuous, Discrete, CompositeVar, Configuration, integrate

pytestmark = pytest.mark.gpu
PI = math.pi


def check(result, expect, ratio=7.0):
    mean = np.concatenate([np.atleast_1d(m) for m in result.mean])
    err = np.concatenate([np.atleast_1d(e) for e in result.stdev])
    expect = np.atleast_1d(np.asarray(expect, dtype=float))
    for ei in range(len(expect)):
        assert abs(mean[ei] - expect[ei]) < err[ei] * ratio, (mean, err, expect)


def Sphere1(neval, alg):
    X = Continuous(0.0, 1.0)
    return integrate("return (x[0]*x[0] + x[1]*x[1] < 1.0) ? 1.0 : 0.0;", var=(X,), dof=[[2]], neval=neval, print=-1, solver=alg, seed=101)


def Sphere2(totalstep, alg, offset=0):
    T = Continuous(0.0, 1.0, 2 + offset, offset=offset)  # small pool: resized implicitly (configuration.jl:156-160)
    config = Configuration(var=(T,), dof=[[2], [3]], neighbor=[(1, 3), (1, 2)], seed=102 + offset)
    return integrate(mci.catalog.sphere2(), config=config, neval=totalstep, print=-1, solver=alg, debug=True)


def TestDiscrete(totalstep, alg):
    X = Discrete(1, 3, adapt=True)
    config = Configuration(var=(X,), dof=[[1]], seed=103)
    return integrate("return x[0];", config=config, neval=totalstep, niter=10, print=-1, solver=alg, debug=True)


def TestDiscrete2(totalstep, alg):
    X = Discrete([(1, 3), (1, 4)], adapt=True)
    config = Configuration(var=(X,), dof=[[1]], seed=104)
    return integrate("return 1.0;", config=config, neval=totalstep, niter=10, print=-1, solver=alg, debug=True)


def TestSingular1(totalstep, alg):
    return integrate("return log(x[0]) / sqrt(x[0]);", neval=totalstep, print=-1, solver=alg, seed=105)


def TestSingular2(totalstep, alg):
    return integrate("return 1.0 / (1.0 - cos(x[0]) * cos(x[1]) * cos(x[2])) / (M_PI*M_PI*M_PI);", var=(Continuous(0.0, PI),),
                     dof=[[3]], neval=totalstep, print=-1, solver=alg, seed=106)


def TestSingular2_CompositeVar(totalstep, alg):
    C = CompositeVar(Continuous(0.0, PI), Continuous(0.0, PI), Continuous(0.0, PI))
    return integrate(mci.catalog.singular2(), var=C, dof=1, neval=totalstep, print=-1, solver=alg, seed=107)


def TestSingular2_Continuous_HighDim(totalstep, alg):
    C = Continuous([(0.0, PI), (0.0, PI), (0.0, PI)])
    return integrate(mci.catalog.singular2(), var=C, dof=1, neval=totalstep, print=-1, solver=alg, seed=108)


def TestHyperSphere(totalstep, alg, N):
    return integrate(mci.catalog.hypersphere(N), var=Continuous(-1, 1), dof=[[i + 1] for i in range(1, N + 1)], userdata=N,
                     neval=totalstep, print=-1, solver=alg, seed=109)


@pytest.mark.parametrize("alg,neval", [("vegas", 200000), ("vegasmc", 100000), ("mcmc", 200000)])  # test/montecarlo.jl:262-387
def test_battery(alg, neval):
    check(Sphere1(neval, alg), PI / 4.0)
    check(Sphere2(neval, alg), [PI / 4.0, 4.0 * PI / 3.0 / 8])
    check(Sphere2(neval, alg, offset=2), [PI / 4.0, 4.0 * PI / 3.0 / 8])
    check(TestDiscrete(neval, alg), 6.0)
    check(TestDiscrete2(neval, alg), 12.0)
    res = TestSingular1(neval, alg)
    check(res, -4.0)
    if alg != "mcmc":
        assert res.stdev[0] < (0.0004 if alg == "vegas" else 0.0007)  # test/montecarlo.jl:317, :364
    check(TestSingular2(neval, alg), 1.3932)
    check(TestSingular2_CompositeVar(neval, alg), 1.3932)
    check(TestSingular2_Continuous_HighDim(neval, alg), 1.3932)
    check(TestHyperSphere(neval, alg, 3), [0.9230, 0.94724, 0.96118])


def TestComplex1(totalstep, alg):
    # f(x, c) = x[1] + x[1]^2 * 1im   (test/montecarlo.jl:166-170)
    return integrate("w[0] = x[0]; w[1] = x[0] * x[0];", neval=totalstep, print=-1, type=complex, solver=alg, debug=True, seed=110)


def TestComplex2(totalstep, alg):
    # integrand returns (x[1], x[1]^2 * 1im): real -> complex conversion  (test/montecarlo.jl:172-185)
    return integrate("w[0] = x[0]; w[1] = 0.0; w[2] = 0.0; w[3] = x[0] * x[0];", dof=[[1], [1]], neval=totalstep, print=-1, type=complex,
                     solver=alg, debug=True, seed=111)


def Sphere3(totalstep, alg, offset=0):
    # two integrands, nested observable shape, user measure  (test/montecarlo.jl:53-92)
    measure = mci.Measure("""
        if (idx < 0 || idx == 0) obs_add(0, rw[0]);                                  // obs[1] += relativeWeights[1]
        if (idx < 0 || idx == 1) { obs_add(1, rw[1]); obs_add(2, rw[1] * 2.0); }     // obs[2][1], obs[2][2]
    """)
    T = Continuous(0.0, 1.0, offset=offset)
    config = Configuration(var=(T,), dof=[[2], [3]], neighbor=[(1, 3), (1, 2)], obs=[0.0, [0.0, 0.0]], seed=112)
    return integrate(mci.catalog.sphere2(), config=config, neval=totalstep, print=-1, solver=alg, debug=True, measure=measure)


def check_complex(result, expect, ratio=7.0):
    # test/runtests.jl:17-29
    expect = np.atleast_1d(np.asarray(expect, dtype=complex))
    for ei in range(len(expect)):
        m, e = complex(np.ravel(result.mean[ei])[0]), complex(np.ravel(result.stdev[ei])[0])
        assert abs(m.real - expect[ei].real) < e.real * ratio, (result.mean, result.stdev, expect)
        assert abs(m.imag - expect[ei].imag) < e.imag * ratio, (result.mean, result.stdev, expect)


@pytest.mark.parametrize("alg,neval", [("vegas", 200000), ("vegasmc", 100000), ("mcmc", 200000)])
def test_complex_and_user_measure(alg, neval):
    check_complex(TestComplex1(neval, alg), 0.5 + 1j / 3)                            # test/montecarlo.jl:292, :331, :376
    check_complex(TestComplex2(neval, alg), [0.5, 1j / 3])
    res = Sphere3(neval, alg)                                                        # test/montecarlo.jl:273
    assert abs(res.mean[0] - PI / 4) < 7 * res.stdev[0]
    np.testing.assert_array_less(np.abs(res.mean[1] - np.array([PI / 6, PI / 3])), 7 * res.stdev[1])


def test_readme_example_and_report(capsys):
    # README.md:26-27: -4.000214 +- 0.000300 with the default solver, neval=1e5
    res = integrate("return log(x[0]) / sqrt(x[0]);", neval=1e5, seed=5, print=0)
    assert abs(res.mean[0] + 4.0) < 7 * res.stdev[0] and res.stdev[0] < 7e-4
    out = capsys.readouterr().out
    assert "wgt average" in out and "ignore" in out
    # src/main.jl:64-65 docstring example
    res = integrate("return x[0]*x[0] + x[1]*x[1];", var=Continuous(0.0, 1.0), dof=[[2]], verbose=-2, solver="vegas", seed=6)
    check(res, 2.0 / 3.0)


def test_interface_accepts_tuple_dof_and_unknown_kwargs():
    # test/interface_tests.jl:1-6
    res = integrate("return 1.0;", dof=[(1,)], vars=Continuous(0, 1), seed=7)
    check(res, 1.0)


def test_bubble_with_resume():
    # test/bubble.jl:93-133: Lindhard polarisation at 4 q; the second call resumes from the trained config
    p = mci.catalog.bubble_parameters()
    sys_tol = {"vegas": 20.0, "vegasmc": 10.0, "mcmc": 10.0}  # test/bubble.jl:124, :131-133
    from catalog_params import bubble_exact
    exact = bubble_exact()
    for alg in ("vegas", "vegasmc", "mcmc"):
        T = Continuous(0.0, p["beta"], alpha=3.0, adapt=True)
        R = Continuous(0.0, 1.0, alpha=3.0, adapt=True)
        theta = Continuous(0.0, PI, alpha=3.0, adapt=True)
        phi = Continuous(0.0, 2 * PI, alpha=3.0, adapt=True)
        Ext = Discrete(1, 4, adapt=False)
        kw = dict(measure=mci.bin_by(4), var=(R, theta, phi, T, Ext), dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], solver=alg)
        result = integrate(mci.catalog.bubble(), neval=1e5, print=-1, block=8, seed=11, **kw)
        result = integrate(mci.catalog.bubble(), neval=1e6, print=-1, block=64, niter=1, config=result.config, solver=alg,
                           measure=mci.bin_by(4))
        avg, std = result.mean[0], result.stdev[0]
        for idx in range(4):
            assert abs(avg[idx] - exact[idx]) < sys_tol[alg] * std[idx], (alg, avg, std, exact)


def test_mcmc_constant_integrand_with_reweight_goal_and_many_chains():
    # test/montecarlo.jl:16 (reweight_goal) ; many chains per block vs the reference's single chain
    res = integrate("return 1.0;", var=(Continuous(0.0, 1.0),), dof=[[1]], neval=1e5, print=-1, solver="mcmc", reweight_goal=np.ones(2), seed=31)
    check(res, 1.0)
    one = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), dof=[[2], [3]], neval=4e5, solver="mcmc", nchain=1, seed=32)
    many = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), dof=[[2], [3]], neval=1e8, solver="mcmc", seed=33)
    check(one, [PI / 4.0, 4.0 * PI / 3.0 / 8])
    check(many, [PI / 4.0, 4.0 * PI / 3.0 / 8], ratio=5.0)   # 1e8 steps: a burn-in bias of 1e-4 would show here
    assert np.all(np.asarray(many.stdev) < 0.1 * np.asarray(one.stdev))


def test_resume_keeps_trained_grid_and_improves_first_iteration():
    # docs/src/index.md:129-150
    res0 = integrate("return log(x[0]) / sqrt(x[0]);", solver="vegas", neval=1e5, seed=21)
    g0 = res0.config.var[0].grid.copy()
    assert not np.allclose(g0, np.linspace(0, 1, 1000))  # trained
    # `var.histogram` like the reference's: train! ends with clearStatistics! (variable.jl:238, :565), so an adaptive run leaves 1e-10;
    # with adapt = false nobody clears it and the last iteration's sum of (|f| jac)^2 per increment stays (variable.jl:196-200, :208-210)
    h = res0.config.var[0].histogram
    assert h.shape == (999,) and np.array_equal(h, np.full(999, 1e-10))
    resf = integrate("return log(x[0]) / sqrt(x[0]);", solver="vegas", neval=1e5, niter=2, adapt=False, config=res0.config)
    h = resf.config.var[0].histogram
    assert h.shape == (999,) and np.all(h > 1e-10) and h.max() < 30.0 * h.mean()          # (on a trained map the increments carry alike)
    assert np.array_equal(resf.config.var[0].grid, g0)
    res = integrate("return log(x[0]) / sqrt(x[0]);", solver="vegas", neval=1e5, config=res0.config)
    assert res.iter_std[0, 0] < 0.2 * res0.iter_std[0, 0]
    check(res, -4.0)


def test_32_bit_stream_passes_the_reference_battery():
    """rng_bits=32 (opt-in, :vegas): the reference's 7-sigma targets (test/montecarlo.jl:298-318) on a 2^-32 lattice of uniforms"""
    res = integrate("return (x[0]*x[0] + x[1]*x[1] < 1.0) ? 1.0 : 0.0;", var=Continuous(0.0, 1.0), dof=[[2]], neval=2e5, solver="vegas", seed=101, rng_bits=32)
    check(res, PI / 4.0)
    res = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), dof=[[2], [3]], neval=2e5, solver="vegas", seed=102, rng_bits=32)
    check(res, [PI / 4.0, 4.0 * PI / 3.0 / 8])
    res = integrate("return log(x[0]) / sqrt(x[0]);", solver="vegas", neval=2e5, seed=103, rng_bits=32)
    check(res, -4.0)
    assert res.stdev[0] < 4e-4                                                      # test/montecarlo.jl:303 "@test res.stdev[1] < 0.0004"
    L = math.sqrt(50.0)
    res = integrate(mci.catalog.gaussian(16), var=Continuous(-L, L), dof=[[16]], neval=1e8, niter=10, solver="vegas", seed=104, rng_bits=32)
    check(res, math.erf(5.0) ** 16, ratio=5.0)
    assert res.stdev[0] < 2e-5


def test_seven_round_philox_stream_passes_the_reference_battery():
    """rng_rounds=7 (opt-in, every solver): the reference's 7-sigma targets (test/montecarlo.jl:262-387) and the 16-D Gaussian at 1e9
    samples on Philox4x32-7"""
    for alg, neval in (("vegas", 200000), ("vegasmc", 100000), ("mcmc", 200000)):
        res = integrate("return (x[0]*x[0] + x[1]*x[1] < 1.0) ? 1.0 : 0.0;", var=Continuous(0.0, 1.0), dof=[[2]], neval=neval, solver=alg, seed=201, rng_rounds=7)
        check(res, PI / 4.0)
        res = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), dof=[[2], [3]], neval=neval, solver=alg, seed=202, rng_rounds=7)
        check(res, [PI / 4.0, 4.0 * PI / 3.0 / 8])
        res = integrate("return log(x[0]) / sqrt(x[0]);", solver=alg, neval=neval, seed=203, rng_rounds=7)
        check(res, -4.0)
        res = integrate("return x[0];", var=Discrete(1, 3, adapt=True), dof=[[1]], neval=neval, solver=alg, seed=204, rng_rounds=7)
        check(res, 6.0)
    L = math.sqrt(50.0)
    res = integrate(mci.catalog.gaussian(16), var=Continuous(-L, L), dof=[[16]], neval=1e8, niter=10, solver="vegas", seed=205, rng_rounds=7)
    assert abs(res.mean[0] - math.erf(5.0) ** 16) < 5.0 * res.stdev[0] and res.stdev[0] < 2e-5


def test_trained_variables_carry_over_into_a_new_configuration():
    """`integrate(...; var = (res.config.var[1], ...))` (docs/src/index.md:129): a variable object keeps what train! taught it, so a NEW
    Configuration built from it -- here even for another integrand and another dof -- starts from the trained grid and distribution, and
    the variables read back the state of the engine they now live in."""
    v0, d0 = Continuous(0.0, 1.0), Discrete(1, 4)
    res0 = integrate("w[0] = log(x[0]) / sqrt(x[0]) * x[1];", var=(v0, d0), dof=[[1, 1]], solver="vegas", neval=1e5, seed=21)
    g0, p0 = res0.config.var[0].grid.copy(), res0.config.var[1].distribution.copy()
    assert not np.allclose(g0, np.linspace(0, 1, 1000)) and not np.allclose(p0, 0.25)
    cfg = Configuration(var=(res0.config.var[0], res0.config.var[1]), dof=[[1, 1]], seed=22)
    eng = mci.Engine(cfg, mci.Integrand("w[0] = log(x[0]) / sqrt(x[0]) * x[1];"))
    np.testing.assert_array_equal(eng.grid(0), g0)                      # seeded from the trained state, not from the uniform grid
    np.testing.assert_allclose(eng.distribution(1)[0], p0, rtol=1e-15)
    np.testing.assert_array_equal(cfg.var[0].grid, g0)                  # ... and the variable now reads the new engine
    res = integrate("w[0] = log(x[0]) / sqrt(x[0]) * x[1];", var=(res0.config.var[0], res0.config.var[1]), dof=[[1, 1]], solver="vegas", neval=1e5, seed=23)
    assert res.iter_std[0, 0] < 0.3 * res0.iter_std[0, 0]              # first iteration already on the trained map
    check(res, -40.0)                                                   # -4 x (1 + 2 + 3 + 4)


def test_a_new_integrand_on_a_trained_configuration_keeps_grid_and_reweight():
    """integrate(g; config = res.config) with another integrand: the problem is rebuilt, the trained grids AND the learned reweight
    (configuration.jl:50 lives across integrate calls) move over, the old engine is released."""
    res0 = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), dof=[[2], [3]], solver="vegasmc", neval=2e5, seed=5)
    cfg = res0.config
    old = cfg._engine
    g0, rw0 = old.grid(0).copy(), old.reweight().copy()
    assert not np.allclose(rw0, 1.0 / 3.0)
    res = integrate("w[0] = (x[0] * x[0] + x[1] * x[1] < 1.0) ? 2.0 : 0.0; w[1] = (x[0] * x[0] + x[1] * x[1] + x[2] * x[2] < 1.0) ? 2.0 : 0.0;",
                    config=cfg, solver="vegasmc", neval=2e5, niter=1, adapt=False, seed=6)
    assert cfg._engine is not old and old.p is None                     # closed, not left to the garbage collector
    np.testing.assert_array_equal(cfg._engine.grid(0), g0)
    check(res, [2 * PI / 4.0, 2 * 4.0 * PI / 3.0 / 8], ratio=6.0)


def test_large_grids_train_in_one_cu(oracle):
    """ninc = 4000: train! stages (5 * nbin + 16) doubles of one variable in LDS -- above 64 KiB the kernels need the dynamic-LDS
    attribute; beyond what 160 KiB hold the Configuration is refused with a message (not a HIP error at the first train!)."""
    cfg = Configuration(var=Continuous(0.0, 1.0, ninc=4000), dof=[[2]], seed=9)
    eng = mci.Engine(cfg, mci.catalog.x2y2())
    ocfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0, npts=4000)], [[2]])
    eng.set_train_walk("serial")
    eng.run("vegas", 20000, 0, 8, 0, 9)
    eng.finish("vegas", 8, adapt=True)
    ocfg.iteration(oracle.VEGAS, "x2y2", None, 20000, 0, 8, 0, 9)
    ocfg.train()
    g = eng.grid(0)
    assert len(g) == 4000 and np.all(np.diff(g) > 0)
    np.testing.assert_allclose(g, ocfg.grid(0), rtol=0, atol=1e-12)
    res = integrate(mci.catalog.x2y2(), var=Continuous(0.0, 1.0, ninc=4000), dof=[[2]], solver="vegas", neval=2e5, seed=10)
    check(res, 2.0 / 3.0)
    with pytest.raises(mci.MCIError) as e:
        mci.Engine(Configuration(var=Continuous(0.0, 1.0, ninc=5000), dof=[[1]]), "return x[0];")
    assert "increments" in str(e.value)


def test_rccl_comm_refuses_an_engine_on_another_device():
    """an RcclComm belongs to one device's context; an engine elsewhere would skip the all-reduce silently"""
    from mcintegration_jl_amd.comm import RcclComm
    comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
    assert comm.library_ranks() == (0, 1)

    class Elsewhere:
        device = 1

        def reduce(self):
            raise AssertionError("must not be reached")
    with pytest.raises(RuntimeError) as e:
        comm.all_reduce(Elsewhere())
    assert "device" in str(e.value)


def test_library_rccl_single_rank_and_torch_reducer():
    """RCCL inside the library with nranks=1 (API use, stream ordering) and the torch reducer leave the
    packed buffer unchanged for a single rank."""
    from mcintegration_jl_amd.comm import RcclComm
    cfg = Configuration(var=Continuous(0.0, 1.0), dof=[[2]], seed=3)
    eng = mci.Engine(cfg, mci.catalog.x2y2())
    eng.run("vegas", 5000, 0, 4, 0, 3)
    before = eng.get_packed()
    comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
    comm.all_reduce(eng)
    np.testing.assert_array_equal(eng.get_packed(), before)
    # mci_comm_sum (the lineage sums of a run of carried chains go through it): a few host doubles through the library's communicator
    np.testing.assert_array_equal(comm.sum_host(eng, np.array([1.5, -2.0, 3.25e-7])), [1.5, -2.0, 3.25e-7])
    res = integrate(mci.catalog.x2y2(), var=Continuous(0.0, 1.0), dof=[[2]], solver="vegas", neval=1e5, seed=4, comm=comm)
    check(res, 2.0 / 3.0)
    # the chain solvers through the N > 1 code path (all-reduce of `visited`, then doReweight! on every rank): same numbers as alone
    for alg in ("vegasmc", "mcmc"):
        kw = dict(dof=[[2], [3]], solver=alg, neval=2e5, seed=5)
        # (a fresh variable per run: a variable object carries its trained grid into the next Configuration, like the reference's)
        a = integrate(mci.catalog.sphere2(), comm=comm, var=Continuous(0.0, 1.0), **kw)
        b = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), **kw)
        np.testing.assert_allclose(a.mean, b.mean, rtol=1e-6)
        assert a.correlated == b.correlated
        np.testing.assert_allclose(a._flat_std, b._flat_std, rtol=1e-4)      # the lineage error, per iteration through the communicator | inside the library
        check(a, [PI / 4.0, 4.0 * PI / 3.0 / 8])
    # adapt = false: every iteration counts, and the loop over run / reduce / finish tells its launches so (mci_set_iteration_counted) the way
    # mci_integrate does on its own -- same chain counts (the first launch on the never-refined map: 64 floors instead of 8), same numbers
    kw = dict(dof=[[2], [3]], solver="vegasmc", neval=2e6, seed=5, adapt=False, niter=4)
    a = integrate(mci.catalog.sphere2(), comm=comm, var=Continuous(0.0, 1.0), **kw)
    b = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), **kw)
    np.testing.assert_allclose(a.mean, b.mean, rtol=1e-6)
    assert a.config._engine.last_chain_launch() == b.config._engine.last_chain_launch() and a.config._engine.last_chain_launch()[1]
    check(a, [PI / 4.0, 4.0 * PI / 3.0 / 8])
    # ONE collective per iteration whatever the solver (BASELINE north star): the 64 holding-time counts an :mcmc launch with automatic
    # chain counts measures ride in the packed all-reduce (exact doubles behind the tables) instead of in an all-reduce of their own
    for alg, extra in (("vegas", 0), ("vegasmc", 0), ("mcmc", 64)):
        cfg = Configuration(var=Continuous(0.0, 1.0), dof=[[2], [3]], seed=6)
        n0 = eng.comm_collectives()[0]
        r = integrate(mci.catalog.sphere2(), config=cfg, comm=comm, solver=alg, neval=2e5, niter=6)
        n1, count = cfg._engine.comm_collectives()
        assert n1 - n0 == 6 + r.warmup, (alg, n1 - n0, r.warmup)
        assert count == cfg._engine.packed_size + extra, (alg, count, cfg._engine.packed_size)


def test_torch_nccl_reducer_works_on_the_device_buffer_in_place():
    """TorchDistComm on a cuda device: the engine's packed buffer is wrapped zero-copy and all-reduced by torch's RCCL
    process group on the library's stream (the bench's fallback when the library cannot bootstrap RCCL itself)."""
    import os
    import torch
    import torch.distributed as dist
    from mcintegration_jl_amd.comm import TorchDistComm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["MCI_TEST_MADE_PROCESS_GROUP"] = "1"   # (conftest.pytest_unconfigure: this process leaves through os._exit)
    try:
        comm = TorchDistComm(tensor_device="cuda:0")
        cfg = Configuration(var=Continuous(0.0, 1.0), dof=[[2]], seed=3)
        eng = mci.Engine(cfg, mci.catalog.x2y2())
        eng.run("vegas", 5000, 0, 4, 0, 3)
        before = eng.get_packed()
        comm.all_reduce(eng)
        np.testing.assert_array_equal(eng.get_packed(), before)
        assert comm._view[1].data_ptr() == eng.packed_device_ptr()          # zero copy
        res = integrate(mci.catalog.x2y2(), var=Continuous(0.0, 1.0), dof=[[2]], solver="vegas", neval=1e5, seed=4, comm=comm)
        ref = integrate(mci.catalog.x2y2(), var=Continuous(0.0, 1.0), dof=[[2]], solver="vegas", neval=1e5, seed=4)
        np.testing.assert_allclose(res.mean[0], ref.mean[0], rtol=1e-12)     # a 1-rank sum changes nothing
    finally:
        dist.destroy_process_group()


def test_outer_threads_run_independent_integrations_concurrently():
    """test/thread.jl:1-38 "outer Threads": host threads call integrate at the same time, each with its own variables and
    configuration (they share the process's device context and stream); every (power, solver) pair must come out right and
    equal to the same call made alone."""
    import threading
    out, errs = {}, []

    def one(i, alg):
        body = "return %s;" % "*".join(["x[0]"] * i)
        return integrate(body, var=(Continuous(0.0, 1.0),), dof=[[1]], print=-1, solver=alg, seed=500 + i, neval=2e5 if alg != "mcmc" else 1e5)

    def work(i):
        try:
            for alg in ("vegas", "vegasmc", "mcmc"):
                out[(i, alg)] = one(i, alg)
        except Exception as e:  # pragma: no cover
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in (1, 2, 3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for (i, alg), res in out.items():
        check(res, 1.0 / (1 + i))
        alone = one(i, alg)
        np.testing.assert_allclose(res.mean[0], alone.mean[0], rtol=1e-9 if alg == "vegas" else 1e-6)


def test_state_file_round_trip_resumes_in_a_new_problem(tmp_path):
    """trained grids / distributions / reweight survive a process boundary through an MCISTATE file;
    a fresh Configuration that loads it starts as well trained as `config=res.config` does (docs/src/index.md:129)."""
    path = tmp_path / "state.mcistate"
    res0 = integrate("return log(x[0]) / sqrt(x[0]);", solver="vegas", neval=1e5, seed=21)
    res0.config.save(path)
    fresh = Configuration(seed=22).load(path)
    res = integrate("return log(x[0]) / sqrt(x[0]);", solver="vegas", neval=1e5, config=fresh)
    np.testing.assert_array_equal(fresh.var[0].grid[[0, -1]], [0.0, 1.0])
    assert res.iter_std[0, 0] < 0.2 * res0.iter_std[0, 0]     # first iteration already runs on the trained grid
    check(res, -4.0)
    # vegasmc state: reweight and a Discrete distribution
    cfg = Configuration(var=(Continuous(0.0, 1.0), Discrete(1, 3)), dof=[[1, 1]], seed=23)
    integrate("return x[0] * x[1];", config=cfg, solver="vegasmc", neval=1e5)
    cfg.save(path)
    cfg2 = Configuration(var=(Continuous(0.0, 1.0), Discrete(1, 3)), dof=[[1, 1]], seed=24)
    eng2 = mci.Engine(cfg2, "return x[0] * x[1];")
    eng2.load_state(path)
    np.testing.assert_array_equal(eng2.grid(0), cfg._engine.grid(0))
    np.testing.assert_allclose(eng2.distribution(1)[0], cfg._engine.distribution(1)[0], rtol=1e-15)   # re-normalised on load (variable.jl:312)
    np.testing.assert_allclose(eng2.reweight(), cfg._engine.reweight(), rtol=1e-15)
    with pytest.raises(mci.MCIError):   # a file for a different problem is refused
        mci.Engine(Configuration(var=Continuous(0.0, 1.0), dof=[[2]]), "return x[0];").load_state(path)


def test_report_config_prints_the_acceptance_table(capsys):
    """report(config) (configuration.jl:345-464): one ChangeIntegrand row per edge of the neighbor graph, one ChangeVariable
    and one SwapVariable row per (integrand, variable), with the numbers of config.propose / config.accept (their parity with
    the oracle is in test_hip_parity.py)."""
    import re
    res = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), dof=[[2], [3]], solver="mcmc", neval=2e5, seed=41)
    pr, ac = res.config._engine.acceptance()
    assert pr.shape == (3, 3, 3) and np.all(ac <= pr)
    assert np.array_equal(res.config.propose, pr) and np.array_equal(res.config.accept, ac)      # the reference's field names (configuration.jl:58-59)
    mci.report(res.config)
    out = capsys.readouterr().out
    for word in ("Configuration", "ChangeIntegrand", "ChangeVariable", "SwapVariable", "Visited", "ReWeight", "Integrand evaluation"):
        assert word in out
    # default graph for N = 2 (configuration.jl:203-208): Norm -> 1; 1 -> Norm, 1 -> 2; 2 -> 1
    rows = re.findall(r"^(Norm -> +\d+|  \d ->Norm|  \d -> +\d+): +([0-9.]+)% +([0-9.]+)% +([0-9.]+)$", out, flags=re.M)
    assert [r[0].split() for r in rows] == [["Norm", "->", "1"], ["1", "->Norm"], ["1", "->", "2"], ["2", "->", "1"]], rows
    neval = res.config.neval
    for (label, p_, a_, ratio), (i, j) in zip(rows, [(2, 0), (0, 2), (0, 1), (1, 0)]):
        assert float(p_) == pytest.approx(pr[0, i, j] / neval * 100.0, abs=1e-6)
        assert float(a_) == pytest.approx(ac[0, i, j] / neval * 100.0, abs=1e-6)
        assert float(ratio) == pytest.approx(ac[0, i, j] / pr[0, i, j], abs=1e-6)
    var_rows = re.findall(r"^  +(\d) / Continuous +: +([0-9.]+)% +([0-9.]+)% +([0-9.]+)$", out, flags=re.M)
    assert [r[0] for r in var_rows] == ["1", "2", "1", "2"]                         # ChangeVariable 1, 2 then SwapVariable 1, 2
    assert float(var_rows[0][1]) == pytest.approx(pr[1, 0, 0] / neval * 100.0, abs=1e-6)
    assert float(var_rows[3][2]) == pytest.approx(ac[2, 1, 0] / neval * 100.0, abs=1e-6)
    res = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), dof=[[2], [3]], solver="vegasmc", neval=1e5, seed=42)
    pr, ac = res.config._engine.acceptance()
    assert pr[1, 0, 0] > 1.0 and 0.0 < ac[1, 0, 0] <= pr[1, 0, 0] and pr[0].max() < 1e-6 and pr[2].max() < 1e-6
    mci.report(res.config)
    assert "ChangeVariable" in capsys.readouterr().out


@pytest.fixture(params=["host", "traced"])
def closure_path(request, monkeypatch):
    """Python closures reach the kernels two ways: as host batch callbacks (trace=False) or written out as device source by the tracer
    (the default).  The closure-vs-device-source tests below run under both."""
    import sys
    I = sys.modules[integrate.__module__]          # (the module: the package exports the function under the same name)
    monkeypatch.setattr(I, "TRACE_DEFAULT", False if request.param == "host" else None)
    return request.param


def test_python_closure_as_measure_matches_device_source(closure_path):
    """a host `measure` closure (mci_set_measure_host; the reference's Sphere3 measure, test/montecarlo.jl:71-84) against the same
    measure as device source: same draws, same relative weights, so the block observables agree to summation-order rounding --
    also with measurefreq and next to a host integrand"""
    def sphere3_measure(x, obs, weights, config):          # measure(vars, obs, weights, config)
        obs[0][0] += weights[0].sum()
        obs[1][0] += weights[1].sum()
        obs[1][1] += (weights[1] * 2.0).sum()
    dev = mci.Measure("obs_add(0, rw[0]); obs_add(1, rw[1]); obs_add(2, rw[1] * 2.0);")
    kw = dict(dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], solver="vegas", neval=2e5, niter=6, seed=77)
    for mf in (1, 3):
        a = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), measure=sphere3_measure, measurefreq=mf, **kw)
        b = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), measure=dev, measurefreq=mf, **kw)
        np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9)
        np.testing.assert_allclose(a.iter_std, b.iter_std, rtol=1e-6)
        assert a.mean[1][1] == pytest.approx(2.0 * a.mean[1][0], rel=1e-9)
    check(a, [PI / 4.0, 4.0 * PI / 3.0 / 8])
    # host integrand AND host measure: everything user-side on the host, draws and statistics on the device
    c = integrate(lambda x, cfg: ((x[0] ** 2 + x[1] ** 2 < 1.0) * 1.0, (x[0] ** 2 + x[1] ** 2 + x[2] ** 2 < 1.0) * 1.0),
                  var=Continuous(0.0, 1.0), measure=sphere3_measure, measurefreq=3, **kw)
    np.testing.assert_allclose(c.iter_mean, a.iter_mean, rtol=1e-9)


def test_python_closure_as_measure_under_the_chain_solvers_matches_device_source(closure_path):
    """:vegasmc and :mcmc call `measure` inside the step loop (vegas_mc/montecarlo.jl:224-227, mcmc/montecarlo.jl:166-169).  With a
    host closure every chain leaves its measured configurations and relative weights in its block's record and the closure runs
    over them after the launch (mci_set_measure_host / _indexed): same chains, same relative weights as the same measure given as
    device source -- the reference's chain (one per block) and many chains, measurefreq, the four-argument and the :mcmc
    five-argument form, a measure that reads the configuration, host integrand AND host measure together."""
    def sphere3_measure(x, obs, weights, config):          # measure(vars, obs, weights, config)
        obs[0][0] += weights[0].sum()
        obs[1][0] += weights[1].sum()
        obs[1][1] += (weights[1] * 2.0).sum()
    def sphere3_measure5(idx, x, obs, weight, config):     # measure(idx, vars, obs, weight, config); idx is 0-based here
        if idx == 0:
            obs[0][0] += weight.sum()
        else:
            obs[1][0] += weight.sum()
            obs[1][1] += (weight * 2.0).sum()
    dev = mci.Measure("obs_add(0, rw[0]); obs_add(1, rw[1]); obs_add(2, rw[1] * 2.0);")
    f2 = lambda X, c: ((X[0] ** 2 + X[1] ** 2 < 1.0) * 1.0, (X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0) * 1.0)
    for solver in ("vegasmc", "mcmc"):
        for mf, nchain, block in ((1, 16, 16), (3, 1, 4)):
            kw = dict(dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], solver=solver, neval=2e4, niter=3, seed=77, nchain=nchain, block=block, measurefreq=mf)
            b = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), measure=dev, **kw)
            for m in (sphere3_measure, sphere3_measure5):   # (measure_form: either form under either solver -- the engine's extension; by default the solver decides)
                a = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), measure=m, measure_form="indexed" if m is sphere3_measure5 else "plain", **kw)
                np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9, err_msg="%s mf=%d nchain=%d %s" % (solver, mf, nchain, m.__name__))
                np.testing.assert_allclose(a.iter_std, b.iter_std, rtol=1e-6)
                assert a.mean[1][1] == pytest.approx(2.0 * a.mean[1][0], rel=1e-9)
        # everything user-side on the host: one launch + one integrand callback per Markov step, the measure after the launch
        kw = dict(dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], solver=solver, neval=8e3, niter=2, seed=5, nchain=16, measurefreq=2)
        c = integrate(f2, var=Continuous(0.0, 1.0), measure=sphere3_measure, integrand_form="plain", measure_form="plain", **kw)
        d = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), measure=dev, **kw)
        np.testing.assert_allclose(c.iter_mean, d.iter_mean, rtol=1e-9)
    # a measure that reads the configuration: histogram of x[0] in two bins, weight of the one integrand (several pools, a Discrete)
    def binned(v, obs, weights, config):
        lo = v[0][0] < 0.5
        obs[0][0] += weights[0][lo].sum()
        obs[0][1] += weights[0][~lo].sum()
    devb = mci.Measure("obs_add(x[0] < 0.5 ? 0 : 1, rw[0]);")
    for solver in ("vegasmc", "mcmc", "vegas"):
        kw = dict(dof=[[1, 1]], obs=[[0.0, 0.0]], solver=solver, neval=2e4, niter=3, seed=9)
        a = integrate("return x[0] * x[1];", var=(Continuous(0.0, 1.0), Discrete(1, 3)), measure=binned, measure_form="plain", **kw)
        b = integrate("return x[0] * x[1];", var=(Continuous(0.0, 1.0), Discrete(1, 3)), measure=devb, **kw)
        np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9)
    assert a.mean[0][0] == pytest.approx(0.75, abs=5 * a.stdev[0][0] + 1e-3) and a.mean[0][1] == pytest.approx(2.25, abs=5 * a.stdev[0][1] + 1e-3)


def test_kernel_timing_events_follow_the_launch_size():
    """the HIP events around a sample launch (mci_kernel_times_ms) cost ~11 us per iteration: launches below 2^20 samples run
    without them unless asked (mci_set_kernel_timing)"""
    cfg = mci.Configuration(var=Continuous(0.0, 1.0), dof=[[2]])
    eng = mci.Engine(cfg, mci.catalog.x2y2())
    eng.integrate("vegas", neval=10**4, niter=4, block=16, seed=1)
    assert len(eng.kernel_times_ms(4)[0]) == 0
    eng.set_kernel_timing(1)
    eng.integrate("vegas", neval=10**4, niter=4, block=16, seed=1, first_iteration=4)
    ms = eng.kernel_times_ms(4)[0]
    assert len(ms) == 4 and (ms > 0).all() and (ms < 1.0).all()
    eng.set_kernel_timing(-1)
    eng.integrate("vegas", neval=2 * 10**6, niter=3, block=16, seed=1, first_iteration=8)
    assert len(eng.kernel_times_ms(3)[0]) == 3


def test_host_closures_refuse_launches_whose_records_do_not_fit():
    """every record of a host closure crosses PCIe into pinned memory: a launch of more than 8 GiB of them is refused with a
    message instead of exhausting the host"""
    with pytest.raises(mci.MCIError) as e:
        integrate("return x[0];", measure=lambda x, obs, w, c: None, dof=[[16]], solver="vegas", neval=2e8, niter=1, seed=1, trace=False)
    assert "8 GiB" in str(e.value)
    with pytest.raises(mci.MCIError) as e:
        integrate(lambda x, c: x[0], dof=[[16]], solver="vegas", neval=2e8, niter=1, seed=1, trace=False)
    assert "8 GiB" in str(e.value)


def test_python_closure_as_integrand_matches_device_source(closure_path):
    """a host closure through the batch-callback path (mci_set_integrand_host) sees the same draws as
    the device-source integrand, so the two runs agree to libm rounding; it reads like the reference's README call."""
    res_h = integrate(lambda x, c: np.log(x[0]) / np.sqrt(x[0]), solver="vegas", neval=1e5, seed=5)     # README.md:26
    res_d = integrate("return log(x[0]) / sqrt(x[0]);", solver="vegas", neval=1e5, seed=5)
    np.testing.assert_allclose(res_h.iter_mean, res_d.iter_mean, rtol=1e-7)
    check(res_h, -4.0)
    # two integrands with different dof on one pool (Sphere2), a tuple return value
    f = lambda X, c: ((X[0] ** 2 + X[1] ** 2 < 1.0) * 1.0, (X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0) * 1.0)
    res = integrate(f, var=Continuous(0.0, 1.0), dof=[[2], [3]], solver="vegas", neval=2e5, seed=6)
    check(res, [PI / 4.0, 4.0 * PI / 3.0 / 8])
    # several variable types: x is a tuple of per-pool arrays; complex output
    g = lambda v, c: v[0][0] * v[1][0] + 1j * v[0][0]
    res = integrate(g, var=(Continuous(0.0, 1.0), Discrete(1, 3)), dof=[[1, 1]], type=complex, solver="vegas", neval=1e5, seed=7)
    check_complex(res, 3.0 + 1.5j)


def test_python_closure_under_mcmc_matches_device_source(closure_path):
    """:mcmc calls `integrand(idx, var, config)` inside every Markov step (mcmc/updates.jl:35-38).  A host closure -- the
    reference's three-argument form, or the two-argument form returning every integrand -- runs one launch and one batch
    callback per step (mcmc_host_step) on the same streams and arithmetic as mcmc_chains: the same chains."""
    kw = dict(dof=[[2], [3]], neval=4e4, niter=3, seed=13, nchain=16, solver="mcmc")
    def f3(idx, X, c):          # reference form; idx is 0-based here
        r2 = X[0] ** 2 + X[1] ** 2 + (X[2] ** 2 if idx == 1 else 0.0)
        return (r2 < 1.0) * 1.0
    f2 = lambda X, c: ((X[0] ** 2 + X[1] ** 2 < 1.0) * 1.0, (X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0) * 1.0)
    d = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), **kw)
    for f in (f3, f2):   # (f3 is what :mcmc calls; the tuple form runs under it when asked for: integrand_form)
        a = integrate(f, var=Continuous(0.0, 1.0), integrand_form="indexed" if f is f3 else "plain", **kw)
        np.testing.assert_allclose(a.iter_mean, d.iter_mean, rtol=1e-9)
        np.testing.assert_allclose(a.iter_std, d.iter_std, rtol=1e-6)
        pa, aa = a.config._engine.acceptance()
        pd_, ad = d.config._engine.acceptance()
        np.testing.assert_allclose(pa, pd_, rtol=1e-12)
        np.testing.assert_allclose(aa, ad, rtol=1e-12)
    # the reference's chain (one per block), several pools with a Discrete, a start that has to be redrawn for some chains
    # (the indicator is zero on 21 % of the square), measurefreq
    kw = dict(var=(Continuous(0.0, 1.0), Discrete(1, 3)), dof=[[2, 1]], neval=3e3, niter=2, seed=4, nchain=1, block=4, solver="mcmc", measurefreq=2)
    h = integrate(lambda idx, v, c: (v[0][0] ** 2 + v[0][1] ** 2 < 1.0) * v[1][0], **kw)
    kw["var"] = (Continuous(0.0, 1.0), Discrete(1, 3))
    ds = integrate("return (x[0] * x[0] + x[1] * x[1] < 1.0) ? x[2] : 0.0;", **kw)
    np.testing.assert_allclose(h.iter_mean, ds.iter_mean, rtol=1e-9)
    # the three-argument form under the other solvers: the library asks it for every integrand in turn
    v = integrate(f3, var=Continuous(0.0, 1.0), dof=[[2], [3]], neval=1e5, solver="vegas", seed=2, integrand_form="indexed")
    check(v, [PI / 4.0, 4.0 * PI / 3.0 / 8])


def _volume_inverse(d):          # test/montecarlo.jl:205-208
    return (d / (2 * PI * math.e)) ** (d / 2) * math.sqrt(d) * math.sqrt(PI)


@pytest.mark.parametrize("alg,neval", [("vegas", 200000), ("vegasmc", 100000)])
def test_inplace_closures_of_the_references_battery(alg, neval, closure_path):
    """The reference's own two in-place battery members as CLOSURES, called as integrate(f; ..., inplace=true) calls them
    (main.jl:26, vegas/montecarlo.jl:140-141, vegas_mc/updates.jl:67-70): TestComplex2_inplace (test/montecarlo.jl:187-196; run at :330,
    :380) and TestHyperSphere (:204-216; run at :333, :383) -- traced into the kernel and on the host batch-callback path, each equal to
    its device-source twin iteration by iteration (same draws, same arithmetic: 1e-9) and inside the reference's 7 sigma."""
    def integrand(x, f, c):                        # test/montecarlo.jl:188-192 (0-based)
        f[0] = x[0]
        f[1] = x[0] ** 2 * 1j
    kw = dict(dof=[[1], [1]], neval=neval, print=-1, type=complex, solver=alg, debug=True, seed=111)
    res = integrate(integrand, inplace=True, **kw)
    twin = integrate("w[0] = x[0]; w[1] = 0.0; w[2] = 0.0; w[3] = x[0] * x[0];", **kw)
    assert isinstance(res.config._engine.integrand, mci.Integrand if closure_path == "traced" else mci.HostIntegrand)
    np.testing.assert_allclose(res.iter_mean, twin.iter_mean, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.iter_std, twin.iter_std, rtol=1e-6, atol=1e-12)
    check_complex(res, [0.5, 1j / 3])

    def f_ternary(x, w, c):                        # test/montecarlo.jl:210-216, ternary and all: the tracer writes the branch out as a select
        _w = x[0] ** 2
        for i in range(c.userdata):
            _w += x[i + 1] ** 2
            w[i] = _volume_inverse(i + 2) if _w < 1.0 else 0.0

    def f_where(x, w, c):                          # ... the batch spelling for the host path (the ternary runs there too, sample by sample: below)
        _w = x[0] ** 2
        for i in range(c.userdata):
            _w = _w + x[i + 1] ** 2
            w[i] = np.where(_w < 1.0, _volume_inverse(i + 2), 0.0)
    f = f_ternary if closure_path == "traced" else f_where
    N = 3
    kw = dict(dof=[[i + 2] for i in range(N)], neval=neval, print=-1, solver=alg, debug=False, seed=18)
    res = integrate(f, var=Continuous(-1, 1), userdata=N, inplace=True, **kw)
    twin = integrate(mci.catalog.hypersphere(N), var=Continuous(-1, 1), **kw)
    np.testing.assert_allclose(res.iter_mean, twin.iter_mean, rtol=1e-9)
    np.testing.assert_allclose(res.iter_std, twin.iter_std, rtol=1e-6)
    check(res, [0.9230, 0.94724, 0.96118])
    # the flag decides, not the parameter count: the same closures without inplace=true are a TypeError, not another form
    with pytest.raises(TypeError, match="inplace"):
        integrate(f, var=Continuous(-1, 1), userdata=N, **kw)
    if closure_path == "host":   # the ternary spelling on the host path: numpy refuses the truth value of a batch, the trampoline goes sample by sample
        small = dict(kw, neval=4000, niter=3)
        a = integrate(f_ternary, var=Continuous(-1, 1), userdata=N, inplace=True, **small)
        b = integrate(mci.catalog.hypersphere(N), var=Continuous(-1, 1), **small)
        assert isinstance(a.config._engine.integrand, mci.HostIntegrand)
        np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9)


def test_python_closure_under_the_default_solver_matches_device_source(closure_path):
    """the reference's default solver is :vegasmc (main.jl:72) and calls the closure inside every Markov step
    (vegas_mc/updates.jl:67-75).  With a host closure the chains of a launch advance in lock step, one kernel launch per step
    (vegasmc_host_step): same Philox streams, same arithmetic -> the same chains as the same function given as device source."""
    kw = dict(dof=[[2], [3]], neval=4e4, niter=4, seed=11, nchain=16)
    f = lambda X, c: ((X[0] ** 2 + X[1] ** 2 < 1.0) * 1.0, (X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0) * 1.0)
    a = integrate(f, var=Continuous(0.0, 1.0), **kw)        # solver defaults to vegasmc  (a fresh variable each: a trained one carries over)
    b = integrate(mci.catalog.sphere2(), var=Continuous(0.0, 1.0), **kw)
    np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9)
    np.testing.assert_allclose(a.iter_std, b.iter_std, rtol=1e-6)
    pa, aa = a.config._engine.acceptance()
    pb, ab = b.config._engine.acceptance()
    np.testing.assert_allclose(pa, pb, rtol=1e-12)
    np.testing.assert_allclose(aa, ab, rtol=1e-12)
    np.testing.assert_allclose(a.config.var[0].grid, b.config.var[0].grid, rtol=0, atol=1e-9)
    # the README call with the default solver and the reference's chain (one per block), smooth integrand, measurefreq
    h = integrate(lambda x, c: x[0] ** 2 + x[1] ** 2, dof=[[2]], neval=2e4, niter=3, seed=3, nchain=1, block=16, measurefreq=2)
    d = integrate("return x[0] * x[0] + x[1] * x[1];", dof=[[2]], neval=2e4, niter=3, seed=3, nchain=1, block=16, measurefreq=2)
    np.testing.assert_allclose(h.iter_mean, d.iter_mean, rtol=1e-9)
    # several pools (the wave-shared pool pick), a Discrete among them, automatic chain count
    g = lambda v, c: v[0][0] * v[1][0]
    hs = integrate(g, var=(Continuous(0.0, 1.0), Discrete(1, 3)), dof=[[1, 1]], neval=2e4, niter=3, seed=7)
    ds = integrate("return x[0] * x[1];", var=(Continuous(0.0, 1.0), Discrete(1, 3)), dof=[[1, 1]], neval=2e4, niter=3, seed=7)
    np.testing.assert_allclose(hs.iter_mean, ds.iter_mean, rtol=1e-9)
    check(hs, 3.0)


@pytest.mark.parametrize("alg", ["vegas", "vegasmc"])
def test_cuba11_against_the_references_printed_results(alg):
    """example/benchmark/cuba/benchmark.jl:119-158 -- the one workload the reference prints results AND a wall time for
    (0.246 s :vegas, 0.495 s :vegasmc at neval=1e5 x 10): same call, means within 5 combined sigma of the printed ones."""
    import json
    import os
    import time
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cuba11_printed.json")) as fh:
        ref = json.load(fh)
    integrate(mci.catalog.cuba11(), dof=[[3]] * 11, neval=1e4, solver=alg, print=-1, seed=50)     # compile first, like benchmark.jl:138
    t0 = time.perf_counter()
    res = integrate(mci.catalog.cuba11(), dof=[[3]] * 11, neval=1e5, solver=alg, print=-1, seed=51)
    dt = time.perf_counter() - t0
    printed = ref["mcintegration_%s_1e5x10" % alg]
    for other in (printed, ref["cuba_vegas_1e6"]):
        for k in range(11):
            assert abs(res.mean[k] - other["mean"][k]) < 5.0 * math.hypot(res.stdev[k], other["sigma"][k]), (alg, k, res.mean[k], other["mean"][k])
    assert dt < printed["wall_seconds"], (dt, printed["wall_seconds"])   # unspecified CPU vs MI355X; a sanity bound, not a benchmark
    print("cuba11 %s: %.4f s (reference prints %.3f s)" % (alg, dt, printed["wall_seconds"]))


@pytest.mark.parametrize("alg", ["vegas", "vegasmc"])
def test_cuba11_as_the_references_inplace_closure(alg):
    """example/benchmark/cuba/benchmark.jl:35-88, :138-140: the published call is `integrate(test3; dof = [[3,] for i in 1:11], neval = 1e5,
    solver = alg, inplace = true)` with test3(x, f, c) storing t1 .. t11 -- two of them ternaries on the draws.  The same closure in
    Python, traced (in-place form, branches as selects), against the hand-written body of the catalog: iteration by iteration."""
    pi = math.pi
    rsq = lambda x, y, z: x * x + y * y + z * z

    def test3(v, f, c):
        x, y, z = v[0], v[1], v[2]
        f[0] = np.sin(x) * np.cos(y) * np.exp(z)
        f[1] = 1.0 / ((x + y) * (x + y) + 0.003) * np.cos(y) * np.exp(z)
        f[2] = 1.0 / (3.75 - np.cos(pi * x) - np.cos(pi * y) - np.cos(pi * z))
        f[3] = abs(rsq(x, y, z) - 0.125)
        f[4] = np.exp(-rsq(x, y, z))
        f[5] = 1.0 / (1.0 - x * y * z + 1e-10)
        f[6] = np.sqrt(abs(x - y - z))
        f[7] = np.exp(-x * y * z)
        f[8] = x * x / (np.cos(x + y + z + 1.0) + 5.0)
        f[9] = 1.0 / np.sqrt(x * y * z + 1e-5) if x > 0.5 else np.sqrt(x * y * z)
        f[10] = 1.0 if rsq(x, y, z) < 1.0 else 0.0
    kw = dict(dof=[[3]] * 11, neval=1e5, solver=alg, print=-1, seed=51)
    a = integrate(test3, inplace=True, trace=True, **kw)
    b = integrate(mci.catalog.cuba11(), **kw)
    assert isinstance(a.config._engine.integrand, mci.Integrand) and a.config._engine.integrand.body.count("?") == 2
    np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-8)
    np.testing.assert_allclose(a.mean, b.mean, rtol=1e-8)


def test_plain_c_consumer_of_the_abi():
    """examples/mci_demo.c: the README integral through the C ABI from plain C (no Python / torch types in the path)."""
    import os
    import subprocess
    demo = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "mci_demo")
    out = subprocess.run([demo, "100000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Integral 1 = -4.0" in out.stdout or "Integral 1 = -3.99" in out.stdout, out.stdout


def test_error_bars_have_the_size_the_reference_prints():
    """tests/golden/printed_error_bars.json through the product on the GPU: same neval / niter / solver as the reference's
    printed examples -> within 7 sigma of exact, error bar within a factor 3 of the printed one (geometric mean of 4 seeds)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "printed_error_bars.json")) as fh:
        cases = json.load(fh)["cases"]
    body = dict(log_over_sqrt="return log(x[0]) / sqrt(x[0]);", x2y2="return x[0]*x[0] + x[1]*x[1];")
    for case in cases:
        sig = []
        for seed in (61, 62, 63, 64):
            res = integrate(body[case["integrand"]], var=Continuous(case["lower"], case["upper"], alpha=case["alpha"]), dof=[[case["dof"]]],
                            solver=case["solver"], neval=case["neval"], niter=10, seed=seed, nchain=1)   # nchain=1: the reference's chain
            check(res, case["exact"])
            sig.append(res.stdev[0])
        ratio = float(np.exp(np.mean(np.log(sig)))) / case["printed_sigma"]
        assert 1.0 / 3.0 < ratio < 3.0, (case["name"], sig, case["printed_sigma"])


def test_auto_chain_counts_are_unbiased_on_sticky_integrands():
    """The many-chains-per-block decomposition is this engine's own; its automatic chain count must not trade bias for
    throughput.  Integrands with heavy-tailed |f|/q are the hard case (short chains under-sample the sticky states):
    measured before the policy was made conservative: :mcmc on the bubble 2.7 % = 55 sigma off with 1e3-step chains,
    :vegasmc on 1/(1 - cos x cos y cos z) 6 sigma low with 381-step chains (profiles/r01_chain_bias.txt)."""
    # :vegasmc, singular2, 1e9 steps in the production phase
    cfg = Configuration(var=Continuous(0.0, PI), dof=[[3]], seed=71)
    integrate(mci.catalog.singular2(), config=cfg, solver="vegasmc", neval=1e8, niter=5)
    res = integrate(mci.catalog.singular2(), config=cfg, solver="vegasmc", neval=1e8, niter=10, ignore=0)
    assert abs(res.mean[0] - 1.3932039296856769) < 5 * res.stdev[0] and res.stdev[0] < 2e-4, (res.mean, res.stdev)
    # :mcmc, bubble diagram, bin 4 (q = 1.5 kF) against a high-statistics :vegas run of the same engine
    p = mci.catalog.bubble_parameters()

    def bubble_cfg(seed):
        var = (Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
               Continuous(0.0, p["beta"], alpha=3.0), Discrete(1, 4, adapt=False))
        return Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=seed)
    ref = integrate(mci.catalog.bubble(), config=bubble_cfg(72), solver="vegas", neval=1e8, niter=10, measure=mci.bin_by(4))
    # the :vegas run is itself pinned: at 1e9 samples it resolves the exact finite-temperature polarisation (beta*EF = 25)
    # from the T = 0 closed form the reference's test compares with at 1e5..1e6 samples (test/bubble.jl:24-36; 1.5e-4 apart in bin 4)
    from catalog_params import bubble_exact, bubble_exact_finite_T
    ft, t0 = np.array(bubble_exact_finite_T()), np.array(bubble_exact())
    assert np.all(np.abs(ref.mean[0] - ft) < 4.5 * ref.stdev[0]), (ref.mean[0], ref.stdev[0], ft)
    assert abs(ref.mean[0][3] - t0[3]) > 4.5 * ref.stdev[0][3], (ref.mean[0], ref.stdev[0], t0)
    cfg = bubble_cfg(73)
    integrate(mci.catalog.bubble(), config=cfg, solver="mcmc", neval=3.2e7, niter=4, measure=mci.bin_by(4))
    res = integrate(mci.catalog.bubble(), config=cfg, solver="mcmc", neval=3.2e7, niter=8, ignore=0, measure=mci.bin_by(4))
    dev = (res.mean[0] - ref.mean[0]) / np.hypot(res.stdev[0], ref.stdev[0])
    assert np.all(np.abs(dev) < 5.0), (res.mean[0], res.stdev[0], ref.mean[0], dev)
    assert np.all(res.stdev[0] < 0.01 * np.abs(ref.mean[0]))   # ... and the test can see a 2.7 % bias


def _seed_scatter(make_cfg, f, measure, solver, nseeds, **kw):
    """(means [seed, obs], reported errors [seed, obs], how many runs were `correlated`) of cold integrate() calls over seeds"""
    ms, es, ncorr = [], [], 0
    for seed in range(1, nseeds + 1):
        res = integrate(f, config=make_cfg(seed), measure=measure, solver=solver, **kw)
        ms.append(res._flat_mean)
        es.append(res._flat_std)
        ncorr += bool(res.correlated)
        res.config._engine.close()
    return np.array(ms), np.array(es), ncorr


def test_error_bars_of_carried_mcmc_chains_are_as_honest_as_the_references_own():
    """BASELINE's metric includes "MC sigma".  The automatic many-chain :mcmc path continues its chains from iteration to iteration, so
    consecutive iterations are not independent, which statistics.jl:186-220 assumes; such a run reports the block-lineage error
    (mci_lineage_sums: blocks never exchange chains).  Seed scatter of the final estimate over the reported error, 128 cold
    integrate() calls each, BASELINE configs[4] and the bubble diagram: within [0.85, 1.15] on average over a configuration's
    observables -- at block = 64.  At the default block = 16 the reference's OWN estimator sits at ~1.14 for independent Gaussian
    iterations (tests/test_lib_host.py test_inverse_variance_weights_from_16_blocks...; measured here on :vegas, which has no chains at
    all: 1.07-1.20, profiles/r04_mcmc_policy.txt), because the weights 1/sigma_i^2 come from 16 blocks each; 64 blocks leave 1.02."""
    nseeds = 128
    kw = dict(neval=1e6, niter=10, block=64)
    c5 = lambda seed: Configuration(var=Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=seed)
    p = mci.catalog.bubble_parameters()

    def bub(seed):
        var = (Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
               Continuous(0.0, p["beta"], alpha=3.0), Discrete(1, 4, adapt=False))
        return Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=seed)
    out = {}
    for name, mk, f, meas in (("c5", c5, mci.catalog.nested_gauss(), None), ("bubble", bub, mci.catalog.bubble(), mci.bin_by(4))):
        for solver in ("mcmc", "vegas"):
            ms, es, ncorr = _seed_scatter(mk, f, meas, solver, nseeds, **kw)
            ratio = ms.std(0, ddof=1) / np.sqrt((es ** 2).mean(0))
            out[(name, solver)] = ratio
            assert 0.85 < ratio.mean() < 1.15 and np.all((ratio > 0.75) & (ratio < 1.30)), (name, solver, ratio)
            if solver == "mcmc" and name == "c5":
                assert ncorr == nseeds       # light tails: many chains per block, carried -> the lineage error is what was reported
    # ... and the chains' error bars are no less honest than those of the solver without chains
    for name in ("c5", "bubble"):
        assert out[(name, "mcmc")].mean() < out[(name, "vegas")].mean() + 0.12, out


def _engine_runs(mk, f, meas, solver, nseeds, neval, block, nchain, niter=10, **kw):
    """(weighted means [seed, obs], reported errors, iteration means [seed, iteration, obs]) of cold mci_integrate calls over seeds"""
    ms, es, im = [], [], []
    for seed in range(1, nseeds + 1):
        eng = mci.Engine(mk(seed), f, measure=meas)
        r = eng.integrate(solver, neval=neval, niter=niter, block=block, seed=seed, nchain=nchain, **kw)
        ms.append(r["mean"].copy())
        es.append(r["stdev"].copy())
        im.append(r["iter_mean"].copy())
        eng.close()
    return np.array(ms), np.array(es), np.array(im)


@pytest.mark.parametrize("solver", ["vegasmc", "mcmc"])
def test_automatic_chains_carry_the_bias_of_the_references_own_chain_and_no_more(solver):
    """BIAS, not scatter.  A block estimate of a chain solver is a ratio of two sums over one correlated chain (main.jl:275-287), and the
    reference's OWN chain (nchain = 1) comes out high at small neval per block: on BASELINE configs[4] at (neval = 1e6, block = 16), 256
    cold runs, +0.12 .. +0.39 sigma per run under :vegasmc and +0.37 .. +0.51 under :mcmc (profiles/r05_bias.txt B; 1.3 .. 1.7 at block =
    64).  The automatic many-chain decomposition must add nothing to that.  Here, 64 seeds of the automatic arm at that same configuration against 32 of the reference chain (:mcmc: 32 against 16):
    (i) the two arms' means agree within 4 standard errors of their difference (seed scatter); (ii) the automatic arm's mean deviation
    per run stays below the reference chain's measured 0.51 sigma + 3 standard errors of a 64-seed mean (0.9 sigma per run: a pooled
    deviation of 7.2 sigma -- the estimator's own; a decomposition that doubled it would fail); (iii) the plain mean of the counted
    iterations, which weights the early iterations of a cold call as much as the late ones, agrees between the arms as well -- that is
    where chains carried across a refinement of the map showed before they were resampled to the moved target (DESIGN "Chains")."""
    # seeds per arm (reference chain, automatic chains).  :mcmc: the reference arm -- 16 sequential chains of 62500 steps -- takes 1.8 s per
    # run, so it gets 16 seeds and the automatic arm 32: bound (ii), which is about the automatic arm, is 0.51 + 3 / sqrt(32) = 1.04
    nseeds = {"reference": 32, "automatic": 64} if solver == "vegasmc" else {"reference": 16, "automatic": 32}
    exact = np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
    c5 = lambda seed: Configuration(var=Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=seed)
    f = mci.catalog.nested_gauss()
    arms = {}
    for label, nchain in (("reference", 1), ("automatic", 0)):
        ms, es, im = _engine_runs(c5, f, None, solver, nseeds[label], 10**6, 16, nchain)
        arms[label] = (ms, es, im[:, 1:].mean(1))
    (ma, ea, ua), (mb, eb, ub) = arms["reference"], arms["automatic"]
    na, nb = nseeds["reference"], nseeds["automatic"]
    diff = (mb.mean(0) - ma.mean(0)) / np.sqrt(ma.var(0, ddof=1) / na + mb.var(0, ddof=1) / nb)
    assert np.all(np.abs(diff) < 4.0), diff
    per_run = (mb.mean(0) - exact) / np.sqrt((eb ** 2).mean(0))
    assert np.all(np.abs(per_run) < 0.51 + 3.0 / math.sqrt(nb)), per_run       # (0.89 at 64 seeds, 1.04 at 32)
    udiff = (ub.mean(0) - ua.mean(0)) / np.sqrt(ua.var(0, ddof=1) / na + ub.var(0, ddof=1) / nb)
    assert np.all(np.abs(udiff) < 4.0), udiff


@pytest.mark.parametrize("name", ["c5", "bubble"])
def test_cold_vegasmc_calls_at_full_size_are_unbiased_iteration_by_iteration(name):
    """integrate(solver = :vegasmc, neval = 1e8, niter = 10) on a FRESH problem, BASELINE configs[2] and the :vegasmc run of configs[4],
    64 seeds: the iterations right behind the first refinements of the map are where carried chains used to be off by 8 .. 17 sigma per
    run-iteration (the target density of a :vegasmc chain contains the map; the stored chains were a sample of the old one:
    profiles/r05_bias.txt A).  With the stored chains resampled to the moved target every counted iteration's mean over the seeds sits
    within 4 standard errors of the exact value and the pooled final estimate within 3.3 -- bounds at what was MEASURED (largest
    per-iteration residue 2.8, pooled 2.5: profiles/r05_bias.txt A4-final, r06_bias.txt), so that a regression that doubles a residue fails."""
    from catalog_params import bubble_exact_finite_T
    nseeds = 64
    if name == "c5":
        mk = lambda seed: Configuration(var=Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=seed)
        f, meas, exact = mci.catalog.nested_gauss(), None, np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
    else:
        p = mci.catalog.bubble_parameters()

        def mk(seed):
            var = (Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
                   Continuous(0.0, p["beta"], alpha=3.0), Discrete(1, 4, adapt=False))
            return Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=seed)
        f, meas, exact = mci.catalog.bubble(), mci.bin_by(4), np.array(bubble_exact_finite_T())
    ms, es, im = _engine_runs(mk, f, meas, "vegasmc", nseeds, 10**8, 16, 0)
    per_iter = (im.mean(0) - exact) / (im.std(0, ddof=1) / math.sqrt(nseeds))
    print("per iteration:\n", np.round(per_iter, 2))
    assert np.all(np.abs(per_iter[1:]) < 4.0), per_iter
    pooled = (ms.mean(0) - exact) / (np.sqrt((es ** 2).sum(0)) / nseeds)
    print("pooled:", np.round(pooled, 2))
    assert np.all(np.abs(pooled) < 3.3), pooled


@pytest.mark.parametrize("name", ["c5", "bubble"])
def test_vegasmc_without_adaptation_counts_every_iteration_without_bias(name):
    """integrate(solver = :vegasmc, adapt = false): every iteration enters the estimate (ignore = 0, main.jl:82) and every launch runs on
    the map as it is -- here the untrained one.  Automatic chains of the usual length have not reached their target there: with every
    iteration started afresh from them the estimate was 3.4 sigma per run low on the 12-D member of BASELINE configs[4] and 2.3 high on
    the bubble diagram's last bin (profiles/r05_bias.txt A6).  The first launch of such a call runs chains 8 x as long, and the
    following ones carry them on (the map has not moved): the mean deviation over 16 seeds stays within 4 of its standard errors
    (1.2 sigma per run) and the last launch is a carried one."""
    from catalog_params import bubble_exact_finite_T
    nseeds = 16
    if name == "c5":
        mk = lambda seed: Configuration(var=Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=seed)
        f, meas, exact = mci.catalog.nested_gauss(), None, np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
    else:
        p = mci.catalog.bubble_parameters()

        def mk(seed):
            var = (Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
                   Continuous(0.0, p["beta"], alpha=3.0), Discrete(1, 4, adapt=False))
            return Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], seed=seed)
        f, meas, exact = mci.catalog.bubble(), mci.bin_by(4), np.array(bubble_exact_finite_T())
    ms, es, im = _engine_runs(mk, f, meas, "vegasmc", nseeds, 10**7, 16, 0, niter=5, adapt=False)
    dev = ((ms - exact) / es).mean(0)
    assert np.all(np.abs(dev) < 1.2), dev
    eng = mci.Engine(mk(1), f, measure=meas)
    eng.integrate("vegasmc", neval=10**7, niter=2, block=16, seed=1, adapt=False)
    assert eng.last_chain_launch()[1] is True
    eng.close()


def test_report_says_when_one_short_chain_per_block_biases_the_estimate():
    """A chain solver's block mean is a ratio of two sums over one chain (main.jl:275-287); with MANY SHORT blocks the ratio estimator's
    bias -- O(tau / N_block), the same in every block -- stands out of an error bar that shrinks with the number of blocks: :mcmc at
    block = 256, neval = 1e6 on BASELINE configs[4] is +5.7 .. +7.3 sigma per run (profiles/r05_odd_calls.txt) with an honest-looking
    error bar.  It is the reference's own chain, so it is reproduced -- and said: Result.chain_bias (statistics.chain_estimator_bias) and a
    note of report(result) that names the reference's knob.  The default call, and the same call at block = 16, carry no note."""
    import io
    exact = np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
    c5 = lambda seed: Configuration(var=Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=seed)
    devs, zs = [], []
    for seed in range(1, 9):
        res = integrate(mci.catalog.nested_gauss(), config=c5(seed), solver="mcmc", neval=1e6, niter=10, block=256)
        assert res.config._engine.last_chain_launch()[0] == 1
        devs.append((res._flat_mean - exact) / res._flat_std)
        zs.append(res.chain_bias["z"])
        out = io.StringIO()
        mci.report(res, io=out)
        assert res.chain_bias["z"] >= 2.0 and "note: solver = :mcmc ran one chain per block of 3906 steps" in out.getvalue() and "fewer blocks" in out.getvalue()
        res.config._engine.close()
    devs = np.array(devs)
    print("block = 256: measured mean deviation per run", np.round(devs.mean(0), 2), " predicted z", np.round(np.mean(zs), 2))
    assert np.all(devs.mean(0) > 2.0)                                            # the bias is there (the reference's own) ...
    assert 0.3 * devs.mean() < np.mean(zs) < 3.0 * devs.mean(), (devs.mean(0), zs)   # ... and the note's estimate is its size
    for kw in (dict(solver="mcmc", neval=1e6, niter=10, block=16), dict(solver="vegasmc", neval=1e6, niter=10, block=256)):
        res = integrate(mci.catalog.nested_gauss(), config=c5(1), **kw)
        out = io.StringIO()
        mci.report(res, io=out)
        assert (res.chain_bias is None or res.chain_bias["z"] < 2.0) and "note: solver" not in out.getvalue(), (kw, res.chain_bias)
        res.config._engine.close()
    for solver in ("vegasmc", "mcmc"):                                           # the reference's default call (main.jl:72-76), README integrand
        res = integrate("return log(x[0]) / sqrt(x[0]);", solver=solver, seed=3)
        out = io.StringIO()
        mci.report(res, io=out)
        assert res.chain_bias is not None and res.chain_bias["z"] < 2.0 and "note: solver" not in out.getvalue(), (solver, res.chain_bias)
        res.config._engine.close()


def test_default_call_of_the_default_solver_is_unbiased_at_its_own_size():
    """The reference's default call is solver = :vegasmc, neval = 1e4, niter = 10, block = 16 (main.jl:72-76): 625 steps per block.  At
    that size a block estimate -- a ratio of two sums over one short chain (main.jl:275-287) -- carries the ratio estimator's
    O(tau / N_block) bias, and more chains per block make it worse (DESIGN "Chains" (iii)); the automatic chain count therefore stays at
    the reference's one chain per block there.  Pinned: over 32 seeds the README integral comes out within 5 sigma of -4 in every run,
    and the mean deviation stays below one sigma per run."""
    devs = []
    for seed in range(1, 33):
        res = integrate("return log(x[0]) / sqrt(x[0]);", seed=seed)       # README.md:26 with every default
        assert res.config._engine.last_chain_launch()[0] == 1
        devs.append((res.mean[0] + 4.0) / res.stdev[0])
        res.config._engine.close()
    devs = np.array(devs)
    assert np.all(np.abs(devs) < 5.0), devs
    assert abs(devs.mean()) < 1.0, (devs.mean(), devs)


def test_bubble_with_fermik_momentum():
    """test/bubble_FermiK.jl:89-124 (part of the reference's runtests.jl): vars = (T, K, Ext) with K = FermiK(3, kF, 0.2 kF,
    10 kF), :mcmc, Steps = 2e5, two calls, every q within 5 sigma of the Lindhard function."""
    from catalog_params import bubble_exact
    p = mci.catalog.bubble_parameters()
    exact = bubble_exact()
    T = Continuous(0.0, p["beta"], alpha=3.0, adapt=True)
    K = mci.FermiK(3, p["kF"], 0.2 * p["kF"], 10.0 * p["kF"])
    Ext = Discrete(1, 4, adapt=False)
    kw = dict(measure=mci.bin_by(2), var=(T, K, Ext), dof=[[1, 1, 1]], obs=[np.zeros(4)], solver="mcmc", neval=2e5, print=-1, block=16)
    result = integrate(mci.catalog.bubble_fermik(), seed=91, **kw)
    result = integrate(mci.catalog.bubble_fermik(), seed=92, **kw)
    avg, std = result.mean[0], result.stdev[0]
    for idx in range(4):
        assert abs(avg[idx] - exact[idx]) < 5.0 * std[idx], (avg, std, exact)


@pytest.mark.parametrize("alg", ["vegas", "vegasmc", "mcmc"])
def test_library_loop_and_per_iteration_loop_give_the_same_run(alg):
    """integrate() hands a single-process run to the library's loop (mci_integrate: iterations queued back to back, one
    read-back); any other communicator goes iteration by iteration through run / all_reduce / finish (main.jl:142-207).
    Same seeds, same iterations, resumed call and reweight goal included: the two agree as two runs of either loop agree -- to
    rounding (the order of the LDS histogram atomics is not fixed, and the grid -> histogram -> grid loop carries that on)."""
    from mcintegration_jl_amd.comm import LocalComm

    class OneRank(LocalComm):   # a communicator type integrate() does not special-case: the per-iteration loop
        pass

    out = []
    for comm in (None, OneRank()):
        cfg = Configuration(var=(Continuous(0.0, 1.0), Discrete(1, 3)), dof=[[2, 1], [3, 0]], seed=77)
        body = "w[0] = x[0] * x[1] * x[3]; w[1] = x[0] + x[1] * x[2];"
        goal = [1.0, 2.0, 1.0] if alg != "vegas" else None
        r1 = integrate(body, config=cfg, solver=alg, neval=2e4, niter=4, comm=comm, reweight_goal=goal)
        r2 = integrate(body, config=cfg, solver=alg, neval=3e4, niter=3, comm=comm, reweight_goal=goal)   # resumes: iterations 4..6
        out.append((r1, r2, cfg._engine.grid(0), cfg._engine.reweight()))
    (a1, a2, ga, wa), (b1, b2, gb, wb) = out
    for a, b in ((a1, b1), (a2, b2)):
        np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-11)
        np.testing.assert_allclose(a.iter_std, b.iter_std, rtol=1e-8)
        np.testing.assert_allclose(a.mean, b.mean, rtol=1e-11)
        np.testing.assert_allclose(a.stdev, b.stdev, rtol=1e-8)
        assert a.neval == b.neval
    np.testing.assert_allclose(ga, gb, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(wa, wb, rtol=1e-11)
