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


def test_a_parameter_sweep_over_a_closure_runs_on_one_code_object():
    """captured floats travel as userdata (trace.py "Captured parameters"): the integral of exp(-a x^2) over [0, 1] for several `a` from
    ONE kernel -- the reference's closures capture their parameters the same way (test/montecarlo.jl:19-92)"""
    objs = []
    for a in (0.5, 2.0, 7.5):
        f = lambda x, c: np.exp(-a * x[0] ** 2)
        r = mci.integrate(f, var=mci.Continuous(0.0, 1.0), dof=[[1]], solver="vegas", neval=100000, niter=6, seed=3, print=-1)
        exact = math.sqrt(math.pi / a) / 2.0 * math.erf(math.sqrt(a))
        assert abs(r.mean[0] - exact) < 5 * r.stdev[0] and r.stdev[0] < 1e-3 * exact, (a, r.mean, r.stdev, exact)
        objs.append(r.config._engine.code_object("vegas"))
    assert objs[0] == objs[1] == objs[2]


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_traced_two_integrand_closure_with_a_select(solver):
    g = lambda x, c: (x[0] ** 2 + x[1] ** 2, mci.trace.where(x[0] > 0.5, x[1], 0.0))
    r = mci.integrate(g, var=mci.Continuous(0.0, 1.0), dof=[[2], [2]], solver=solver, neval=200000, niter=10, seed=7, trace=True, print=-1,
                      integrand_form="plain")      # (:mcmc calls integrand(idx, var, config) by default; the tuple form runs under it when asked for)
    assert abs(r.mean[0] - 2.0 / 3.0) < 5 * r.stdev[0] and abs(r.mean[1] - 0.25) < 5 * r.stdev[1]
    assert r.stdev[0] < 0.02 and r.stdev[1] < 0.02


def test_traced_measure_closures_match_device_source_measures():
    """vegas/montecarlo.jl:156-161, mcmc/montecarlo.jl:166-169: a `measure` closure traced into a device Measure measures what the same
    measure written as device source measures -- four- and five-argument forms, a masked measure next to a Discrete pool, all three
    solvers.  (Every call builds its own variables: a Continuous object carries its trained grid into the next call.)"""
    def sphere3_measure(x, obs, weights, config):
        obs[0][0] += weights[0].sum()
        obs[1][0] += weights[1].sum()
        obs[1][1] += (weights[1] * 2.0).sum()

    def sphere3_measure5(idx, x, obs, weight, config):
        if idx == 0:
            obs[0][0] += weight.sum()
        else:
            obs[1][0] += weight.sum()
            obs[1][1] += (weight * 2.0).sum()
    dev = mci.Measure("obs_add(0, rw[0]); obs_add(1, rw[1]); obs_add(2, rw[1] * 2.0);")
    exact = [math.pi / 4.0, math.pi / 6.0, math.pi / 3.0]
    for solver, m in (("vegas", sphere3_measure), ("vegasmc", sphere3_measure), ("mcmc", sphere3_measure5), ("vegas", sphere3_measure5)):
        # (explicit chain count: the automatic one follows the kernel's workgroup size, which may differ between two measure bodies)
        kw = dict(dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], solver=solver, neval=4e4, niter=6, seed=77, measurefreq=2, print=-1,
                  **({} if solver == "vegas" else dict(nchain=16, block=16)))
        a = mci.integrate(mci.catalog.sphere2(), var=mci.Continuous(0.0, 1.0), measure=m, trace=True,
                          measure_form="indexed" if m is sphere3_measure5 else "plain", **kw)
        b = mci.integrate(mci.catalog.sphere2(), var=mci.Continuous(0.0, 1.0), measure=dev, **kw)
        for r in (a, b):
            got = np.array([r.mean[0], r.mean[1][0], r.mean[1][1]], dtype=np.float64).ravel()
            err = np.array([r.stdev[0], r.stdev[1][0], r.stdev[1][1]], dtype=np.float64).ravel()
            assert np.all(np.abs(got - exact) < 6 * err + 0.01) and np.all(err < 0.2), (solver, m.__name__, got, err)
            assert got[2] == pytest.approx(2.0 * got[1], rel=1e-9)
        if solver == "vegas" and np.allclose(a.iter_mean[0], b.iter_mean[0], rtol=1e-9):      # same samples, same sums: then every iteration agrees
            np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-4, err_msg="%s %s" % (solver, m.__name__))

    def binned(v, obs, weights, config):
        lo = v[0][0] < 0.5
        obs[0][0] += weights[0][lo].sum()
        obs[0][1] += weights[0][~lo].sum()
    for solver in ("vegas", "vegasmc", "mcmc"):
        kw = dict(dof=[[1, 1]], obs=[[0.0, 0.0]], solver=solver, neval=4e4, niter=6, seed=9, print=-1,
                  **({} if solver == "vegas" else dict(nchain=16, block=16)))
        a = mci.integrate("return x[0] * x[1];", var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3)), measure=binned, trace=True, measure_form="plain", **kw)
        got, err = np.ravel(a.mean).astype(np.float64), np.ravel(a.stdev).astype(np.float64)
        assert np.all(np.abs(got - [0.75, 2.25]) < 6 * err + 0.02) and np.all(err < 0.3), (solver, got, err)     # (1 + 2 + 3) * int x dx over [0, .5) | [.5, 1)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc"])
def test_complex_closures_traced_and_on_the_host_agree(solver):
    """Configuration(type = ComplexF64): integrand AND measure as closures -- traced into the kernels (complex values as (re, im) pairs of
    real expressions, trace.CSym) against the same closures on the host callback path: same draws, same arithmetic, iteration by
    iteration; and both inside 7 sigma of the reference's TestComplex2 answers (test/montecarlo.jl:172-185: 1/2 and i/3)"""
    def f(x, c):
        return x[0], x[0] ** 2 * 1j

    def m(x, obs, w, c):
        obs[0][0] += w[0].sum()
        obs[1][0] += w[1].sum()
    out = []
    for trace in (True, False):
        kw = dict(dof=[[1], [1]], type=complex, obs=[0j, 0j], solver=solver, neval=1e5, niter=6, seed=111, print=-1,
                  **({} if solver == "vegas" else dict(nchain=16)))
        r = mci.integrate(f, measure=m, trace=trace, **kw)
        eng = r.config._engine
        assert isinstance(eng.integrand, mci.Integrand if trace else mci.HostIntegrand)
        for k, exact in enumerate((0.5 + 0j, 1j / 3)):
            z, e = complex(np.ravel(r.mean[k])[0]), complex(np.ravel(r.stdev[k])[0])
            assert abs(z.real - exact.real) < 7 * e.real + 1e-12 and abs(z.imag - exact.imag) < 7 * e.imag + 1e-12, (solver, trace, k, z, e)
        out.append(r)
    np.testing.assert_allclose(out[0].iter_mean, out[1].iter_mean, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_the_references_sphere_closure_with_its_ternary(solver):
    """test/montecarlo.jl:19-32: `(x, c) -> x[1]^2 + x[2]^2 < 1.0 ? 1.0 : 0.0` (and `(idx, x, c)` under :mcmc), as a Python closure with the
    same ternary: the tracer runs both ways of the branch and writes a select (trace.explore) -- the device-source twin's iterations to
    1e-9, pi / 4 inside the reference's 7 sigma"""
    if solver == "mcmc":
        f = lambda idx, x, c: 1.0 if x[0] ** 2 + x[1] ** 2 < 1.0 else 0.0
    else:
        f = lambda x, c: 1.0 if x[0] ** 2 + x[1] ** 2 < 1.0 else 0.0
    kw = dict(dof=[[2]], solver=solver, neval=1e5, niter=10, seed=101, print=-1, **({} if solver == "vegas" else dict(nchain=16)))
    a = mci.integrate(f, var=mci.Continuous(0.0, 1.0), trace=True, **kw)
    b = mci.integrate("return (x[0] * x[0] + x[1] * x[1] < 1.0) ? 1.0 : 0.0;", var=mci.Continuous(0.0, 1.0), **kw)
    assert isinstance(a.config._engine.integrand, mci.Integrand) and "? 1.0 : 0.0" in a.config._engine.integrand.body
    np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9)
    assert abs(a.mean[0] - math.pi / 4) < 7 * a.stdev[0]


@pytest.mark.parametrize("solver", ["vegas", "vegasmc"])
def test_random_closures_with_python_control_flow_run_the_same_in_the_kernels_and_on_the_host(solver):
    """The generator of tests/test_trace.py (nested ternaries, early returns, `and` / `or` / `not`, the same test met twice) end to end: each
    closure traced into the kernels (selects) against the same closure kept on the host -- where a truth test on a batch of draws makes
    the trampoline call it sample by sample (engine._make_host_callback) -- iteration by iteration to 1e-9.  Piecewise-polynomial
    integrands with jumps: the map trains on them, the chains step across them."""
    from test_trace import _random_branching_closure
    rng = np.random.default_rng(2027)
    done = 0
    for case in range(6):
        f = _random_branching_closure(rng, 4)
        kw = dict(dof=[[4]], solver=solver, neval=3000, niter=3, seed=300 + case, print=-1, **({} if solver == "vegas" else dict(nchain=8)))
        # (a fresh variable per call: a Continuous carries its trained map into the next Configuration built on it, like the reference's)
        a = mci.integrate(f, var=mci.Continuous(-1.0, 1.0), **kw)       # trace by default; silently the host path where the tracer refuses
        if not isinstance(a.config._engine.integrand, mci.Integrand):
            continue                                       # (more ways through it than the tracer follows: host path on both sides)
        b = mci.integrate(f, var=mci.Continuous(-1.0, 1.0), trace=False, **kw)
        assert isinstance(b.config._engine.integrand, mci.HostIntegrand)
        np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9, atol=1e-12, err_msg="case %d\n%s" % (case, a.config._engine.integrand.body))
        np.testing.assert_allclose(a.iter_std, b.iter_std, rtol=1e-7, atol=1e-12)
        done += 1
    assert done >= 4, done


def test_modulo_floor_division_and_rounding_of_draws():
    """Julia's mod / fld / round on a draw (`%`, `//`, np.round here): written out as fmod with Python's sign convention, floor(a / b),
    rint -- int_0^1 (x mod 0.3 + fld(x, 0.25) + round(4 x)) dx = 0.14 + 1.5 + 2, and the same closure on the host path"""
    f = lambda x, c: x[0] % 0.3 + x[0] // 0.25 + np.round(x[0] * 4)
    kw = dict(solver="vegas", neval=1e5, seed=7, print=-1)
    a = mci.integrate(f, **kw)
    assert isinstance(a.config._engine.integrand, mci.Integrand) and "fmod(" in a.config._engine.integrand.body
    assert abs(a.mean[0] - 3.64) < 5 * a.stdev[0] and a.stdev[0] < 2e-3
    b = mci.integrate(f, trace=False, **kw)
    np.testing.assert_allclose(a.iter_mean, b.iter_mean, rtol=1e-9)
