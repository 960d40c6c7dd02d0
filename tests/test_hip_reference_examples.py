"""What a user of the reference types, as Python closures: every example of README.md:24-75 and docs/src/index.md:33-200, and
test/bubble.jl's integrand and measure (a struct of parameters, a momentum looked up by the Discrete draw, Green's functions with
branches on sampled values, a histogram over the Discrete draw).  All of them are traced into the kernels (asserted); the two histogram
examples also run on the host path (trace=False) and must agree with the traced run iteration by iteration."""
import math
import types

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd import CompositeVar, Configuration, Continuous, Discrete, integrate

from test_trace_tables import GRID, N, histogram_integrand, histogram_measure

pytestmark = pytest.mark.gpu
PI = math.pi


def flat(v):
    return np.concatenate([np.atleast_1d(np.asarray(q, dtype=float)) for q in v])


def check(res, expect, ratio=7.0, traced=True):
    if traced:
        assert isinstance(res.config._engine.integrand, mci.Integrand), "the closure was not written out as device source"
    m, e, v = flat(res.mean), flat(res.stdev), flat(expect)
    assert m.shape == v.shape and np.all(np.abs(m - v) < ratio * e + 1e-12), (m, e, v)
    return res


def test_readme_examples():
    # README.md:24-27
    f = lambda x, c: np.log(x[0]) / np.sqrt(x[0])
    check(integrate(f, var=Continuous(0, 1), neval=1e5, seed=1), [-4.0])
    # :33-37  pi / 4 = 1 - 1/3 + 1/5 - ...: a Discrete variable, the sum to 100 terms
    term = lambda n, c: 4 * ((-1) ** (n[0] + 1)) / (2 * n[0] - 1)
    series = sum(4 * (-1) ** (n + 1) / (2 * n - 1) for n in range(1, 101))
    check(integrate(term, var=Discrete(1, 100), neval=1e5, seed=2), [series])
    # :57-60  a Bool-valued integrand, symmetric variables
    check(integrate(lambda x, c: x[0] ** 2 + x[1] ** 2 < 1, var=Continuous(-1, 1), dof=[[2]], seed=3), [PI])
    # :62-67  a composite variable unpacked into its leaves: g((r, θ), c) = r[1]
    def g(v, c):
        r, theta = v
        return r[0]
    check(integrate(g, var=Continuous([(0, 1), (0, 2 * PI)]), dof=[(1,)], seed=4), [PI])
    check(integrate(g, var=CompositeVar(Continuous(0, 1), Continuous(0, 2 * PI)), dof=[(1,)], seed=4), [PI])
    # :69-74  a tuple of variable vectors: f((n, x), c) = 4 (-1)^n[1] x[1]^(2 n[1])
    def h(v, c):
        n, x = v
        return 4 * (-1) ** n[0] * x[0] ** (2 * n[0])
    check(integrate(h, var=(Discrete(0, 100), Continuous(0, 1)), dof=[(1, 1)], neval=1e5, seed=5), [sum(4 * (-1) ** n / (2 * n + 1) for n in range(101))])


def test_documentation_examples():
    # docs/src/index.md:33  every default but the solver
    check(integrate(lambda x, c: np.log(x[0]) / np.sqrt(x[0]), solver="vegas", seed=6), [-4.0])
    # :74-80  a pool created first; dof = [2, ] -- one integrand per element
    x = Continuous(0.0, 1.0)
    check(integrate(lambda x, c: x[0] ** 2 + x[1] ** 2 < 1.0, var=x, dof=[2, ], seed=7), [PI / 4])
    # :87-91  xy = Continuous([(0, 1), (0, 1)]): ((x, y), c) -> log(x[1]) / sqrt(x[1]) * y[1]
    def f(v, c):
        x, y = v
        return np.log(x[0]) / np.sqrt(x[0]) * y[0]
    check(integrate(f, var=Continuous([(0.0, 1.0), (0.0, 1.0)]), seed=8), [-2.0])
    # :98  a tuple of Bools
    two = lambda X, c: (X[0] ** 2 + X[1] ** 2 < 1.0, X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0)
    check(integrate(two, var=Continuous(0.0, 1.0), dof=[[2], [3]], seed=9), [PI / 4, PI / 6])
    # :106-111  the do-block
    def block(X, c):
        r1 = 1.0 if (X[0] ** 2 + X[1] ** 2 < 1.0) else 0.0
        r2 = 1.0 if (X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0) else 0.0
        return (r1, r2)
    check(integrate(block, var=Continuous(0.0, 1.0), dof=[[2], [3]], seed=10), [PI / 4, PI / 6])
    # :115-119  inplace = true
    def inplace(X, f, c):
        f[0] = 1.0 if (X[0] ** 2 + X[1] ** 2 < 1.0) else 0.0
        f[1] = 1.0 if (X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0) else 0.0
    check(integrate(inplace, var=Continuous(0.0, 1.0), dof=[[2], [3]], inplace=True, seed=11), [PI / 4, PI / 6])
    # :131-134  config = res0.config
    res0 = integrate(lambda x, c: np.log(x[0]) / np.sqrt(x[0]), seed=12)
    res = check(integrate(lambda x, c: np.log(x[0]) / np.sqrt(x[0]), config=res0.config), [-4.0])
    assert res.stdev[0] < res0.stdev[0]


def test_benchmark_scripts():
    """example/benchmark/vegas: benchmark1.jl:28-30 (the do-block calling a function of the pool), benchmark3.jl:31-58 (three integrands
    RETURNED AS A LIST, a loop over the first four draws; Cuba's numbers at :17-19 -- "MCIntegration currently fails" there at neval = 1e4,
    this runs it at 1e5 per iteration)"""
    def f(x):
        return 1.0 / (1.0 - np.cos(x[0]) * np.cos(x[1]) * np.cos(x[2])) / PI ** 3
    check(integrate(lambda var, config: f(var), neval=200000, var=(Continuous(0.0, PI, alpha=3.0, adapt=True),), dof=[[3]], solver="vegas", seed=13), [1.3932], ratio=5.0)

    def f3(x, c):
        dx2 = 0.0
        for d in range(4):
            dx2 += (x[d] - 0.5) ** 2
        f = np.exp(-200 * dx2) * 1000.0
        return [f, f * x[0], f * x[0] ** 2]
    g = 1000.0 * (PI / 200.0) ** 2 * math.erf(math.sqrt(200.0) / 2) ** 4
    res = check(integrate(f3, neval=100000, dof=[[4], [4], [4]], verbose=-1, solver="vegas", seed=14), [g, g / 2, g * (0.25 + 1.0 / 400.0)], ratio=5.0)
    cuba = [0.24681600683822702, 0.12341321349438042, 0.06232499578312799]
    assert np.all(np.abs(flat(res.mean) - cuba) < 5.0 * np.hypot(flat(res.stdev), [0.0003, 0.00014, 7.4e-5]))


@pytest.mark.parametrize("solver", ["vegasmc", "vegas", "mcmc"])
def test_measure_histogram_example(solver):
    """docs/src/index.md "Measure Histogram": the radius looked up in config.userdata by the Discrete draw, `obs[i][bin] += weights[i]`.
    Bin b holds int_0^1 dx [x^2 + r_b^2 < 1] = sqrt(1 - r_b^2) and the quarter disc pi (1 - r_b^2) / 4."""
    r = np.array(GRID)
    exact = [np.sqrt(1 - r ** 2), PI * (1 - r ** 2) / 4]
    if solver == "mcmc":
        f = lambda idx, v, c: histogram_integrand(v, c)[idx]
        m = lambda idx, v, obs, w, c: obs[idx].__setitem__(v[1][0] - 1, obs[idx][v[1][0] - 1] + w)
    else:
        f, m = histogram_integrand, histogram_measure
    kw = dict(measure=m, dof=[[1, 1], [2, 1]], obs=[np.zeros(N), np.zeros(N)], userdata=GRID, neval=1e5, solver=solver, seed=20, print=-1,
              **({} if solver == "vegas" else dict(nchain=16)))
    mk = lambda: (Continuous(0.0, 1.0), Discrete(1, N))
    a = integrate(f, var=mk(), **kw)
    eng = a.config._engine
    assert isinstance(eng.integrand, mci.Integrand) and isinstance(eng.measure, mci.Measure) and "obs_add(" in eng.measure.body
    check(a, exact)
    assert a.stdev[0][N - 1] < 1e-9 and abs(a.mean[0][N - 1]) < 1e-9             # r = 1: nothing inside (up to clearStatistics' 1e-10 offsets)
    small = 2e4 if solver == "vegas" else 4e3               # (a host closure under a chain solver is called once per Markov step)
    b = integrate(f, var=mk(), trace=False, **dict(kw, neval=small))
    assert isinstance(b.config._engine.integrand, mci.HostIntegrand) and isinstance(b.config._engine.measure, mci.HostMeasure)
    a2 = integrate(f, var=mk(), **dict(kw, neval=small))
    np.testing.assert_allclose(flat([flat(q) for q in a2.iter_mean]), flat([flat(q) for q in b.iter_mean]), rtol=1e-9, atol=1e-12)
    # the built-in "bin by a Discrete draw" measure is the same histogram
    c = integrate(f, var=mk(), **dict(kw, neval=small, measure=mci.bin_by(1)))
    np.testing.assert_allclose(flat([flat(q) for q in a2.iter_mean]), flat([flat(q) for q in c.iter_mean]), rtol=1e-9, atol=1e-12)


def _bubble_closures():
    """test/bubble.jl:12-92 in Python: Para, green, integrand, measure (0-based indices; the Discrete draw runs 1 .. Qsize like there)"""
    p = mci.catalog.bubble_parameters()
    para = types.SimpleNamespace(kF=p["kF"], beta=p["beta"], me=p["me"], spin=p["spin"], dim=p["dim"], Qsize=p["Qsize"],
                                 extQ=[np.array([q, 0.0, 0.0]) for q in p["extQ"]])

    def green(tau, omega, beta):
        if tau >= 0.0:
            return np.exp(-omega * tau) / (1 + np.exp(-omega * beta)) if omega > 0.0 else np.exp(omega * (beta - tau)) / (1 + np.exp(omega * beta))
        return -np.exp(-omega * (tau + beta)) / (1 + np.exp(-omega * beta)) if omega > 0.0 else -np.exp(-omega * tau) / (1 + np.exp(omega * beta))

    def integrand(vars, config):
        R, Theta, Phi, T, Ext = vars
        para = config.userdata
        kF, beta, me = para.kF, para.beta, para.me
        r = R[0] / (1 - R[0])
        theta, phi = Theta[0], Phi[0]
        k = np.array([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)])
        factor = 1.0 / (2 * PI) ** para.dim
        factor *= r ** 2 / (1 - R[0]) ** 2 * np.sin(theta)
        Tin, Tout = 0.0, T[0]
        extidx = Ext[0]
        q = para.extQ[extidx - 1]
        kq = k + q
        tau = Tout - Tin
        omega1 = (np.dot(k, k) - kF ** 2) / (2 * me)
        g1 = green(tau, omega1, beta)
        omega2 = (np.dot(kq, kq) - kF ** 2) / (2 * me)
        g2 = green(-tau, omega2, beta)
        n = 0
        return g1 * g2 * para.spin * factor * np.cos(2 * PI * n * tau / beta)

    def measure(vars, obs, weight, config):
        Ext = vars[-1]
        obs[0][Ext[0] - 1] += weight[0]
    return para, integrand, measure


@pytest.mark.parametrize("alg,ratio", [("mcmc", 10.0), ("vegas", 20.0), ("vegasmc", 10.0)])
def test_bubble_as_the_reference_writes_it(alg, ratio):
    """test/bubble.jl:94-133: Steps = 1e5 on block = 8, then 1e6 on block = 64, niter = 1 from the trained configuration; every q
    within `ratio` sigma of the Lindhard function (:124, :131-133)."""
    from catalog_params import bubble_exact
    para, integrand, measure = _bubble_closures()
    f = (lambda idx, v, c: integrand(v, c)) if alg == "mcmc" else integrand
    m = (lambda idx, v, obs, w, c: obs[0].__setitem__(v[-1][0] - 1, obs[0][v[-1][0] - 1] + w)) if alg == "mcmc" else measure
    T = Continuous(0.0, para.beta, alpha=3.0, adapt=True)
    R = Continuous(0.0, 1.0, alpha=3.0, adapt=True)
    theta = Continuous(0.0, PI, alpha=3.0, adapt=True)
    phi = Continuous(0.0, 2 * PI, alpha=3.0, adapt=True)
    Ext = Discrete(1, len(para.extQ), adapt=False)
    kw = dict(measure=m, userdata=para, var=(R, theta, phi, T, Ext), dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(para.Qsize)], solver=alg)
    result = integrate(f, neval=1e5, print=-1, block=8, seed=40, **kw)
    eng = result.config._engine
    assert isinstance(eng.integrand, mci.Integrand) and isinstance(eng.measure, mci.Measure)
    assert eng.integrand.body.count("?") >= 4 and "ud[" in eng.integrand.body and "* (int)fmin(fmax(" in eng.integrand.body
    result = integrate(f, neval=1e6, print=-1, block=64, niter=1, config=result.config, solver=alg, measure=m)
    exact = bubble_exact()
    avg, std = result.mean[0], result.stdev[0]
    for i in range(para.Qsize):
        assert abs(avg[i] - exact[i]) < ratio * std[i], (alg, avg, std, exact)
    # the device-source twin of the catalog on the same seeds: the same estimator, another order of the arithmetic
    cat = integrate(mci.catalog.bubble(), neval=1e5, print=-1, block=8, seed=40,
                    **dict(kw, measure=mci.bin_by(4), var=(Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
                                                            Continuous(0.0, para.beta, alpha=3.0), Discrete(1, 4, adapt=False)), userdata=None))
    first = integrate(f, neval=1e5, print=-1, block=8, seed=40,
                      **dict(kw, var=(Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
                                      Continuous(0.0, para.beta, alpha=3.0), Discrete(1, 4, adapt=False))))
    np.testing.assert_allclose(np.asarray(first.iter_mean[0]).reshape(-1), np.asarray(cat.iter_mean[0]).reshape(-1), rtol=1e-6)


def test_bubble_with_fermik_momentum_as_the_reference_writes_it():
    """test/bubble_FermiK.jl:54-124: vars = (T, K, Ext), `k = K[1]` a momentum VECTOR of the FermiK pool, `kq = k + q` with q a row of
    para.extQ, :mcmc with the five-argument measure; Steps = 2e5, two calls, every q within 5 sigma of the Lindhard function."""
    from catalog_params import bubble_exact
    para, _, _ = _bubble_closures()

    def green(tau, omega, beta):
        if tau >= 0.0:
            return np.exp(-omega * tau) / (1 + np.exp(-omega * beta)) if omega > 0.0 else np.exp(omega * (beta - tau)) / (1 + np.exp(omega * beta))
        return -np.exp(-omega * (tau + beta)) / (1 + np.exp(-omega * beta)) if omega > 0.0 else -np.exp(-omega * tau) / (1 + np.exp(omega * beta))

    def integrand(idx, vars, config):
        T, K, Ext = vars
        para = config.userdata
        kF, beta, me = para.kF, para.beta, para.me
        k = K[0]
        Tin, Tout = 0.0, T[0]
        extidx = Ext[0]
        q = para.extQ[extidx - 1]
        kq = k + q
        tau = Tout - Tin
        omega1 = (np.dot(k, k) - kF ** 2) / (2 * me)
        g1 = green(tau, omega1, beta)
        omega2 = (np.dot(kq, kq) - kF ** 2) / (2 * me)
        g2 = green(-tau, omega2, beta)
        phase = 1.0 / (2 * PI) ** 3
        return g1 * g2 * para.spin * phase

    def measure(idx, vars, obs, weight, config):
        Ext = vars[-1]
        obs[0][Ext[0] - 1] += weight
    T = Continuous(0.0, para.beta, alpha=3.0, adapt=True)
    K = mci.FermiK(3, para.kF, 0.2 * para.kF, 10.0 * para.kF)
    Ext = Discrete(1, len(para.extQ), adapt=False)
    kw = dict(measure=measure, userdata=para, var=(T, K, Ext), dof=[[1, 1, 1]], obs=[np.zeros(para.Qsize)], solver="mcmc", neval=2e5, print=-1, block=16)
    result = integrate(integrand, seed=91, **kw)
    eng = result.config._engine
    assert isinstance(eng.integrand, mci.Integrand) and isinstance(eng.measure, mci.Measure)
    result = integrate(integrand, seed=92, **kw)
    exact = bubble_exact()
    avg, std = result.mean[0], result.stdev[0]
    for i in range(para.Qsize):
        assert abs(avg[i] - exact[i]) < 5.0 * std[i], (avg, std, exact)


@pytest.mark.parametrize("ns,name", [(mci.Vegas, "vegas"), (mci.VegasMC, "vegasmc"), (mci.MCMC, "mcmc")])
def test_driving_a_solver_block_by_block_through_its_seam(ns, name):
    """`Solver.montecarlo(config, integrand, neval, print, timer, debug; measure, measurefreq, inplace)` (src/main.jl:253-264): what
    `_block!` does with it -- m = observable ./ normalization, obsSum += m, obsSquaredSum += m^2 (:275-287), mean and error of the mean
    over the blocks (:296-320) -- by hand, 32 blocks on an untrained map; Sphere2's closures with their measure (test/montecarlo.jl:19-52)."""
    if name == "mcmc":
        f = lambda idx, X, c: 1.0 if X[0] ** 2 + X[1] ** 2 + (X[2] ** 2 if idx == 1 else 0.0) < 1.0 else 0.0

        def measure(idx, X, obs, w, c):
            obs[idx][0] += w
    else:
        f = lambda X, c: (1.0 if X[0] ** 2 + X[1] ** 2 < 1.0 else 0.0, 1.0 if X[0] ** 2 + X[1] ** 2 + X[2] ** 2 < 1.0 else 0.0)

        def measure(X, obs, w, c):
            for i in range(2):
                obs[i][0] += w[i]
    config = Configuration(var=Continuous(0.0, 1.0), dof=[[2], [3]], neighbor=[(1, 3), (1, 2)], seed=60)
    nblock, neval = 32, 20000
    s, s2 = np.zeros(2), np.zeros(2)
    for b in range(nblock):
        out = ns.montecarlo(config, f, neval, 0, [], False, measure=measure, measurefreq=1)
        assert out is config and config.normalization > 0.0
        m = np.array([float(o) for o in config.observable]) / config.normalization
        s += m
        s2 += m * m
    eng = config._engine
    assert isinstance(eng.integrand, mci.Integrand) and isinstance(eng.measure, mci.Measure) and config.iterations_done == nblock
    mean = s / nblock
    err = np.sqrt(np.maximum(0.0, s2 / nblock - mean ** 2) / (nblock - 1))
    exact = np.array([PI / 4, PI / 6])
    assert np.all(np.abs(mean - exact) < 7.0 * err) and np.all(err < 0.01), (mean, err)
    assert config.neval > 0 and config.visited.shape == (3,)
    if name != "vegas":
        assert config.propose.shape == (3, 3, 3) and config.accept.sum() > 0.0
    assert np.all(config.var[0].histogram > 0.0) and config.var[0].histogram.sum() > 1.0     # this block's, nobody has trained on it
    np.testing.assert_allclose(config.var[0].grid, np.linspace(0.0, 1.0, 1000), rtol=0, atol=2e-16)   # the seam never trains (main.jl:190-203 does)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc"])
def test_complex_histogram_by_a_discrete_draw(solver):
    """a ComplexF64 observable binned by the Discrete draw (`obs[1][bin[1]] += weights[1]` with complex weights: every bin is an (re, im)
    pair of slots): bin b holds int_0^1 x dx * exp(i phi_b), the phases looked up in a captured table"""
    K = 6
    phase = np.linspace(0.0, 2.0, K)

    def f(v, c):
        x, b = v
        return x[0] * np.exp(1j * phase[b[0] - 1])

    def measure(v, obs, w, c):
        obs[0][v[1][0] - 1] += w[0]
    kw = dict(measure=measure, dof=[[1, 1]], obs=[np.zeros(K, dtype=complex)], type=complex, solver=solver, neval=1e5, seed=70, print=-1,
              **({} if solver == "vegas" else dict(nchain=16)))
    mk = lambda: (Continuous(0.0, 1.0), Discrete(1, K))
    a = integrate(f, var=mk(), **kw)
    eng = a.config._engine
    assert isinstance(eng.integrand, mci.Integrand) and isinstance(eng.measure, mci.Measure) and "2 * mci_k0_0 + 1" in eng.measure.body
    m, e = np.asarray(a.mean[0]), np.asarray(a.stdev[0])
    exact = 0.5 * np.exp(1j * phase)
    assert np.all(np.abs(m.real - exact.real) < 7 * e.real + 1e-12) and np.all(np.abs(m.imag - exact.imag) < 7 * e.imag + 1e-12), (m, e, exact)
    small = 2e4 if solver == "vegas" else 4e3
    a2 = integrate(f, var=mk(), **dict(kw, neval=small))
    b = integrate(f, var=mk(), trace=False, **dict(kw, neval=small))
    assert isinstance(b.config._engine.measure, mci.HostMeasure)
    np.testing.assert_allclose(np.asarray(a2.iter_mean, dtype=complex).ravel(), np.asarray(b.iter_mean, dtype=complex).ravel(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("solver", ["vegas", "mcmc"])
def test_a_two_dimensional_histogram_over_two_discrete_draws(solver):
    """`obs = [zeros(3, 4)]`, `obs[1][a[1], b[1]] += weights[1]`: an observable with two axes, a vertex table `V[a, b]` looked up by the
    same two draws; bin (a, b) holds V[a, b] / 2.  Traced and on the host, Result.mean with the observable's shape."""
    V = (np.arange(12.0).reshape(3, 4) + 1.0) / 3.0

    def f(v, c):
        x, a, b = v
        return x[0] * V[a[0] - 1, b[0] - 1]

    def measure(v, obs, w, c):
        obs[0][v[1][0] - 1, v[2][0] - 1] += w[0]
    if solver == "mcmc":
        f0, m0 = f, measure
        f = lambda idx, v, c: f0(v, c)
        measure = lambda idx, v, obs, w, c: obs[0].__setitem__((v[1][0] - 1, v[2][0] - 1), obs[0][v[1][0] - 1, v[2][0] - 1] + w)
    kw = dict(measure=measure, dof=[[1, 1, 1]], obs=[np.zeros((3, 4))], solver=solver, neval=1e5, seed=80, print=-1, **({} if solver == "vegas" else dict(nchain=16)))
    mk = lambda: (Continuous(0.0, 1.0), Discrete(1, 3), Discrete(1, 4))
    a = integrate(f, var=mk(), **kw)
    eng = a.config._engine
    assert isinstance(eng.integrand, mci.Integrand) and isinstance(eng.measure, mci.Measure)
    m, e = np.asarray(a.mean[0]), np.asarray(a.stdev[0])
    assert m.shape == (3, 4) and np.all(np.abs(m - V / 2) < 7 * e), (m, e)
    small = 2e4 if solver == "vegas" else 4e3
    a2 = integrate(f, var=mk(), **dict(kw, neval=small))
    b = integrate(f, var=mk(), trace=False, **dict(kw, neval=small))
    assert isinstance(b.config._engine.measure, mci.HostMeasure)
    np.testing.assert_allclose(np.asarray(a2.iter_mean).ravel(), np.asarray(b.iter_mean).ravel(), rtol=1e-9, atol=1e-12)


def test_a_sweep_over_the_bubbles_parameters_runs_on_one_code_object():
    """the struct of parameters in `userdata` travels as trace parameters and a table (trace.py "Captured parameters"): the bubble at three
    temperatures and another set of external momenta from ONE kernel, each inside 20 sigma of its Lindhard values (test/bubble.jl:124)"""
    import types
    from catalog_params import lindhard
    _, integrand, measure = _bubble_closures()
    objs = []
    for beta, qmax in ((25.0, 1.5), (40.0, 1.5), (25.0, 2.5)):
        p = mci.catalog.bubble_parameters(beta=beta)
        para = types.SimpleNamespace(kF=p["kF"], beta=p["beta"], me=p["me"], spin=p["spin"], dim=p["dim"], Qsize=4,
                                     extQ=[np.array([q, 0.0, 0.0]) for q in np.linspace(0.0, qmax * p["kF"], 4)])
        var = (Continuous(0.0, 1.0, alpha=3.0), Continuous(0.0, PI, alpha=3.0), Continuous(0.0, 2 * PI, alpha=3.0),
               Continuous(0.0, para.beta, alpha=3.0), Discrete(1, 4, adapt=False))
        res = integrate(integrand, measure=measure, userdata=para, var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)], solver="vegas",
                        neval=1e6, seed=90, print=-1)
        eng = res.config._engine
        assert isinstance(eng.integrand, mci.Integrand)
        objs.append(eng.code_object("vegas"))
        exact = [lindhard(q[0], p) for q in para.extQ]
        for i in range(4):
            assert abs(res.mean[0][i] - exact[i]) < 20.0 * res.stdev[0][i] + 3e-4, (beta, qmax, res.mean[0], res.stdev[0], exact)   # (+ the finite-T shift)
    assert objs[0] == objs[1] == objs[2]
