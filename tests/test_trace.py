"""Python closures written out as device source (mcintegration_jl_amd/trace.py): the counterpart of Julia inlining `integrand(var, config)`
into the reference's loop (vegas/montecarlo.jl:140-144).  On the CPU: the written-out body is compiled with gcc (the way the oracle
compiles any body) and compared with the closure at random points; it compiles for gfx950 through the library's JIT (offline
context); integrate(..., trace=True) hands an engine the body (here the oracle, end to end); closures that cannot be written out are
refused and take the host callback path."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd.trace import TraceError, arctan2, fmax, fmin, trace_integrand, where


def _erf(x):
    return np.vectorize(math.erf)(x) if isinstance(x, np.ndarray) and x.dtype != object else x.erf() if hasattr(x, "erf") else math.erf(x)


CASES = [
    # name, configuration, closure(s)
    ("gauss4", lambda: mci.Configuration(var=mci.Continuous(-5.0, 5.0), dof=[[4]]),
     lambda x, c: np.exp(-np.sum(x * x) / 2) / (2 * np.pi) ** (len(x) / 2)),
    ("x2y2", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]), lambda x, c: x[0] ** 2 + x[1] ** 2),
    ("two_integrands_two_pools", lambda: mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 6)), dof=[[2, 1], [1, 0]]),
     lambda x, c: (x[0][0] ** 2 + x[0][1] * np.sin(x[1][0]), np.sqrt(x[0][0]) * 2 - 1 / (1 + x[0][0]))),
    ("selects", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3]]),
     lambda x, c: where(x[0] > 0.5, fmax(x[1], 0.2), abs(x[2] - 0.5)) ** 1.5 + fmin(x[0], x[1]) * arctan2(x[1], x[2] + 0.1)),
    ("functions", lambda: mci.Configuration(var=mci.Continuous(0.1, 2.0), dof=[[3]]),
     lambda x, c: np.log(x[0]) * np.cos(x[1]) + np.tanh(x[2]) / np.cosh(x[0]) + np.log1p(x[1]) - np.expm1(-x[2]) + np.arctan(x[0]) + 2.0 ** x[1] + x[2] ** -1 + x[0] ** 0.5 + x[1] ** 5),
    ("sums", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[6]]),
     lambda x, c: sum(x) * np.prod(x[:3] + 1.0) + np.dot(x[::2], x[1::2]) - x[-1] * 3 + 7),
    ("composite", lambda: mci.Configuration(var=mci.CompositeVar(mci.Continuous(0.0, 1.0), mci.Continuous(-1.0, 1.0)), dof=[[2]]),
     lambda x, c: x[0, 0] * x[1, 1] + np.exp(-x[0, 1] ** 2) * x[1, 0]),
    ("numpy_selects_on_single_draws", lambda: mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[3]]),
     lambda x, c: np.maximum(x[0], 0.5) + np.minimum(x[1], x[2]) + np.where(x[0] > 0.2, x[1], -x[2]) * np.sign(x[1]) + np.clip(x[2], -0.3, 0.4)
     + np.hypot(x[0], x[1]) + np.abs(x[2]) + np.heaviside(x[0], 0.5) + np.arctan2(x[1], x[2])),
    ("arrays_with_draws_and_numbers", lambda: mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[3]]),
     lambda x, c: np.sum(2.0 * x) + np.sum(x * x[0]) + (x + x[1])[2] + np.exp(x).sum() + np.sqrt(np.abs(x) + 1.0).prod() + np.power(x[0], 2)
     + np.power(2.0, x[1]) + np.square(x[2]) + (x[0] - x) @ (x / 2.0)),
    ("discrete_equality_and_indicator", lambda: mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3)), dof=[[2, 1], [2, 1]]),
     lambda x, c: (np.where(x[1][0] == 2, x[0][0], 2.0 * x[0][1]) + (x[1][0] != 1) * 0.5, (x[0][0] ** 2 + x[0][1] ** 2 < 1.0) * 1.0)),
    ("comparisons_as_numbers", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]]),
     lambda x, c: (x[0] > 0.5) * 2.0 + (x[1] > 0.5) / 4.0 - (x[0] > x[1]) * 1.0 + (x[0] > 0.2) / ((x[1] > 0.7) + 1.0) + (x[0] < 0.9) / ((x[1] < 2.0) * 1)),
    ("python_branches_on_draws", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [3]]),
     lambda x, c: (1.0 if x[0] ** 2 + x[1] ** 2 < 1.0 else 0.0,                                   # the reference's Sphere ternary (test/montecarlo.jl:19-32)
                   (x[0] if x[1] > 0.5 and x[2] > 0.25 else x[1] * 2.0 if x[0] > 0.5 or x[2] < 0.1 else np.maximum(x, 0.5).sum()))),
    ("modulo_floor_division_rounding", lambda: mci.Configuration(var=mci.Continuous(-2.0, 2.0), dof=[[2], [2]]),        # Julia's mod / fld / round / trunc
     lambda x, c: (x[0] % 0.3 + x[0] % -0.7 + 0.9 % (x[1] + 2.5) + x[0] % (x[1] + 0.1) + np.mod(x[1], 0.4) + np.fmod(x[0], 0.6) + np.remainder(x[1], x[0] + 3.0),
                   x[0] // 0.25 + 1.0 // (x[1] + 2.5) + np.floor_divide(x[1], 0.3) + np.round(x[0] * 4) + round(x[1] * 3) + np.rint(x[0] * 7) + np.trunc(x[1] * 5))),
    ("constant_and_shared_subexpressions", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [2]]),
     lambda x, c: (1.5, np.exp(x[0] * x[1]) + np.exp(x[0] * x[1]) ** 2)),
]


def _c_function(oracle, integrand):
    """the written-out body (an Integrand, or body text) compiled with gcc: fn(x*, w*, ud* = the Integrand's own userdata)"""
    dp = C.POINTER(C.c_double)
    raw = C.CFUNCTYPE(None, dp, dp, dp)(oracle.compile_c_integrand(getattr(integrand, "body", integrand)))
    ud = np.ascontiguousarray(getattr(integrand, "userdata", ()), dtype=np.float64)

    def fn(xp, wp, udp=None):
        raw(xp, wp, udp if udp is not None else ud.ctypes.data_as(dp) if len(ud) else None)
    return fn


@pytest.mark.parametrize("name,cfg,f", CASES, ids=[c[0] for c in CASES])
def test_traced_body_computes_what_the_closure_computes(oracle, name, cfg, f):
    from mcintegration_jl_amd.trace import _argument, _domain_points, _pools
    config = cfg()
    I = trace_integrand(f, config)
    assert isinstance(I, mci.Integrand) and "w[%d] =" % (config.N - 1) in I.body
    fn = _c_function(oracle, I)
    pools, ndraw = _pools(config)
    rng = np.random.default_rng(5)
    X = _domain_points(config, ndraw, 200, rng)
    for p in range(200):
        x = np.ascontiguousarray(X[:, p])
        w = np.zeros(config.N)
        fn(x.ctypes.data_as(C.POINTER(C.c_double)), w.ctypes.data_as(C.POINTER(C.c_double)), None)
        arg = _argument(pools, lambda k: X[k, p])
        arg = tuple(a.astype(np.float64) for a in arg) if isinstance(arg, tuple) else arg.astype(np.float64)
        ref = f(arg, config)
        ref = np.array(ref if isinstance(ref, tuple) else [ref], dtype=np.float64)
        np.testing.assert_allclose(w, ref, rtol=1e-13, atol=1e-300, err_msg="%s at %s\n%s" % (name, x, I.body))
    if name == "constant_and_shared_subexpressions":
        assert I.body.count("exp(") == 1                  # interned: one temporary for the shared subexpression


def test_indexed_form_is_traced_per_integrand(oracle):
    """the reference's :mcmc form integrand(idx, var, config) (mcmc/montecarlo.jl:34-36): one trace per index"""
    config = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1], [2]])
    f = lambda idx, x, c: x[0] if idx == 0 else x[0] * x[1]
    I = trace_integrand(f, config, indexed=True)
    assert "w[0] = x[0];" in I.body and "x[0] * x[1]" in I.body


@pytest.mark.parametrize("what,f", [
    ("math.exp wants a float", lambda x, c: math.exp(x[0])),
    ("a loop that ends on a draw has no bound on its ways", lambda x, c: next(k for k in range(10 ** 6) if k * x[0] > 400.0) * 1.0),
    ("a reduction form of a ufunc on a draw", lambda x, c: np.add.reduce(x[0])),
    ("wrong number of values", lambda x, c: (x[0], x[1])),
    ("not a number", lambda x, c: "one"),
    ("non-finite constant", lambda x, c: x[0] * float("inf")),
])
def test_closures_that_cannot_be_written_out_are_refused(what, f):
    config = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    with pytest.raises(TraceError):
        trace_integrand(f, config)


@pytest.mark.parametrize("what,f", [
    ("numpy adds two comparisons as a logical or, the written-out C adds 1 + 1", lambda x, c: ((x[0] > 0.5) + (x[1] > 0.5)) * 1.0),
    ("numpy multiplies two comparisons as a logical and -- the same number -- but subtracts them as an error", lambda x, c: ((x[0] > 0.5) - (x[1] > 0.5)) * 1.0),
])
def test_numpy_arithmetic_on_comparisons_that_the_body_would_compute_differently_is_refused(what, f):
    """evaluate() -- what the body is checked with -- has the semantics of the emitted C (a comparison is the number 0.0 or 1.0), not
    numpy's (bool + bool is a logical or): where the two differ the closure keeps the host path instead of integrating another function"""
    config = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    with pytest.raises(TraceError):
        trace_integrand(f, config)


def test_captured_floats_become_userdata_slots_and_a_sweep_reuses_one_body(oracle):
    """Captured parameters are not baked into the source: the body of a closure does not depend on their values (one code object in the
    kernel cache serves a parameter sweep), the values travel as userdata; parameter-only subexpressions are evaluated on the host."""
    config = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    scale = np.array([0.5, 2.0])

    def make(a, b):
        return lambda x, c, off=0.25: np.exp(-a * x[0] ** 2) * np.cos(b * np.pi) + np.sum(scale * x) + off / (1.0 + a * a)
    bodies, uds = [], []
    for a, b in ((1.5, 0.25), (3.0, -0.75), (0.1, 2.0)):
        f = make(a, b)
        I = trace_integrand(f, config)
        bodies.append(I.body)
        uds.append(np.array(I.userdata))
        assert "1.5" not in I.body and "ud[" in I.body
        fn = _c_function(oracle, I)
        rng = np.random.default_rng(1)
        for _ in range(50):
            x = rng.uniform(0.0, 1.0, 2)
            w = np.zeros(1)
            fn(x.ctypes.data_as(C.POINTER(C.c_double)), w.ctypes.data_as(C.POINTER(C.c_double)), I.userdata.ctypes.data_as(C.POINTER(C.c_double)))
            assert w[0] == pytest.approx(float(f(x, config)), rel=1e-13)
    assert bodies[0] == bodies[1] == bodies[2] and not np.array_equal(uds[0], uds[1])
    # cos(b * pi) and off / (1 + a * a) are parameter-only: one slot each, no cos() left in the body
    assert "cos(" not in bodies[0] and len(uds[0]) == 5, (bodies[0], uds[0])
    # the same code object: the JIT's cache key is the source
    e1 = mci.Engine(config, trace_integrand(make(1.5, 0.25), config), device=-1)
    e2 = mci.Engine(config, trace_integrand(make(7.0, 1.0), config), device=-1)
    e1.compile("vegas"), e2.compile("vegas")
    assert e1.code_object("vegas") == e2.code_object("vegas")
    e1.close(), e2.close()
    # a captured float the closure BRANCHES on stays a parameter too: both ways are written out and the test's outcome -- parameter-only,
    # evaluated on the host -- travels in a userdata slot (one body for every value; round 5 baked the value in)
    bodies = []
    for thr in (0.5, 0.1):
        g = lambda x, c: x[0] * 2.0 if thr > 0.25 else x[1]
        I = trace_integrand(g, config)
        assert "x[0] * 2.0" in I.body and "(ud[0] != 0.0) ?" in I.body and list(I.userdata) == [1.0 if thr > 0.25 else 0.0]
        bodies.append(I.body)
    assert bodies[0] == bodies[1]
    # captured ints are structure, not data
    n = 2
    I = trace_integrand(lambda x, c: np.sum(x[:n]) ** n, config)
    assert "ud[" not in I.body


def test_negative_zero_is_its_own_constant():
    from mcintegration_jl_amd.trace import _Trace
    t = _Trace()
    assert t.const(0.0) is not t.const(-0.0) and t.const(-0.0) is t.const(-0.0)
    config = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]])
    I = trace_integrand(lambda x, c: x[0] * where(x[0] > 0.5, -0.0, 0.0) + np.arctan2(where(x[0] > 0.5, -0.0, 0.0), -1.0), config)
    assert "(-0.0)" in I.body and ": 0.0" in I.body


def test_a_closure_with_hidden_state_is_refused():
    """the written-out body is checked against the closure itself at random points before it is used"""
    config = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    calls = []

    def f(x, c):
        calls.append(1)
        return x[0] * len(calls)        # a different function every time it is called
    with pytest.raises(TraceError):
        trace_integrand(f, config)
    # (complex weights trace since round 6 -- tests/test_callback_forms.py: a real value in a complex configuration is (value, 0))
    I = trace_integrand(lambda x, c: x[0] * 2, mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]], type=complex))
    assert "w[0] = t" in I.body and "w[1] = 0.0;" in I.body


def test_integrate_with_trace_hands_the_engine_device_source(oracle):
    """integrate(closure, trace=True): the engine gets an Integrand (source), not a host callback -- run here on the oracle end to end
    (same loop as test_distributed_gloo's OracleEngine); a closure that cannot be traced still gets its HostIntegrand"""
    from oracle_engine import OracleEngine
    got = {}

    class Traced(OracleEngine):
        def __init__(self, config, integrand, **kw):
            got["integrand"] = integrand
            if isinstance(integrand, mci.Integrand):
                integrand.name = oracle.compile_c_integrand(integrand.body)
            super().__init__(config, integrand, **kw)

    r = mci.integrate(lambda x, c: x[0] ** 2 + x[1] ** 2, var=mci.Continuous(0.0, 1.0), dof=[[2]], solver="vegas", neval=20000, niter=6,
                      block=8, seed=3, trace=True, engine_factory=Traced, print=-1)
    assert isinstance(got["integrand"], mci.Integrand) and "x[0] * x[0]" in got["integrand"].body
    assert abs(r.mean[0] - 2.0 / 3.0) < 5 * r.stdev[0] and r.stdev[0] < 2e-3

    class Stop(Exception):
        pass

    def factory(config, integrand, **kw):
        got["integrand"] = integrand
        raise Stop()
    for trace in (None, False):
        with pytest.raises(Stop):
            mci.integrate(lambda x, c: math.exp(x[0]), var=mci.Continuous(0.0, 1.0), dof=[[1]], solver="vegas", trace=trace,
                          engine_factory=factory, print=-1)
        assert isinstance(got["integrand"], mci.HostIntegrand)
    with pytest.raises(Stop):   # trace=False: a closure is a host closure
        mci.integrate(lambda x, c: x[0], var=mci.Continuous(0.0, 1.0), dof=[[1]], solver="vegas", engine_factory=factory, print=-1, trace=False)
    assert isinstance(got["integrand"], mci.HostIntegrand)
    with pytest.raises(Stop):   # the default: traced when it can be
        mci.integrate(lambda x, c: x[0], var=mci.Continuous(0.0, 1.0), dof=[[1]], solver="vegas", engine_factory=factory, print=-1)
    assert isinstance(got["integrand"], mci.Integrand)
    with pytest.warns(RuntimeWarning, match="not traced"), pytest.raises(Stop):   # asked for and not possible: the reason is said
        mci.integrate(lambda x, c: math.exp(x[0]), var=mci.Continuous(0.0, 1.0), dof=[[1]], solver="vegas", engine_factory=factory, print=-1, trace=True)


def test_traced_bodies_compile_for_gfx950():
    """every function the tracer can write out exists in the device's math library: the :vegas kernel of each case compiles through
    the library's JIT without a GPU (offline context)"""
    for name, cfg, f in CASES:
        if name in ("composite",):
            continue        # (:vegas has no CompositeVar restriction, but keep the offline compile to the plain layouts)
        config = cfg()
        eng = mci.Engine(config, trace_integrand(f, config), device=-1)
        eng.compile("vegas")
        assert os.path.exists(eng.code_object("vegas")), name
        eng.close()


def _sphere3_measure(x, obs, weights, config):          # the reference's Sphere3 measure (test/montecarlo.jl:71-84), batch-vectorised like a HostMeasure
    obs[0][0] += weights[0].sum()
    obs[1][0] += weights[1].sum()
    obs[1][1] += (weights[1] * 2.0).sum()


def _sphere3_measure5(idx, x, obs, weight, config):     # measure(idx, vars, obs, weight, config), mcmc/montecarlo.jl:166-169; idx 0-based
    if idx == 0:
        obs[0][0] += weight.sum()
    else:
        obs[1][0] += weight.sum()
        obs[1][1] += (weight * 2.0).sum()


def _binned(v, obs, weights, config):
    lo = v[0][0] < 0.5
    obs[0][0] += weights[0][lo].sum()
    obs[0][1] += weights[0][~lo].sum()


def test_measure_closures_are_written_out_as_obs_add_statements():
    """vegas/montecarlo.jl:156-161: measure(vars, obs, relative_weights, config); `obs[i][k] += expr` -> obs_add(flat k, expr)"""
    from mcintegration_jl_amd.trace import trace_measure
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], obs=[0.0, [0.0, 0.0]])
    m = trace_measure(_sphere3_measure, cfg)
    assert isinstance(m, mci.Measure)
    assert [ln.strip() for ln in m.body.splitlines()] == ["const double t7 = rw[1] * 2.0;", "obs_add(0, rw[0]);", "obs_add(1, rw[1]);", "obs_add(2, t7);"] \
        or all(q in m.body for q in ("obs_add(0, rw[0]);", "obs_add(1, rw[1]);", "rw[1] * 2.0"))
    per_sample = lambda x, obs, w, c: (obs[0].__setitem__(0, obs[0][0] + w[0] * x[1]), obs[1].__setitem__(1, obs[1][1] + w[1] * np.exp(-x[0])))
    b = trace_measure(per_sample, cfg).body                   # the reference's own per-sample form
    assert "obs_add(0, " in b and "obs_add(2, " in b and "exp(" in b and "obs_add(1," not in b
    m5 = trace_measure(_sphere3_measure5, cfg, indexed=True).body
    assert "if (idx < 0 || idx == 0) {" in m5 and "if (idx < 0 || idx == 1) {" in m5 and m5.count("obs_add(") == 3
    cfg2 = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3)), dof=[[1, 1]], obs=[[0.0, 0.0]])
    bb = trace_measure(_binned, cfg2).body
    assert "x[0] < 0.5" in bb and "? rw[0] : 0.0" in bb and "!t" in bb
    with pytest.raises(TraceError):                           # a bin chosen by a draw: that is mci.bin_by's job
        trace_measure(lambda x, obs, w, c: obs[0].__setitem__(int(x[0][0] * 2), w[0]), cfg2)
    calls = []
    with pytest.raises(TraceError):                           # not a function of the record
        trace_measure(lambda x, obs, w, c: (calls.append(1), obs[0].__setitem__(0, obs[0][0] + w[0] * len(calls)))[0], cfg2)


def test_integrate_with_trace_hands_the_engine_a_device_measure():
    class Stop(Exception):
        pass
    got = {}

    def factory(config, integrand, measure=None, **kw):
        got["measure"] = measure
        raise Stop()
    kw = dict(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], solver="vegas", engine_factory=factory, print=-1)
    with pytest.raises(Stop):
        mci.integrate(mci.catalog.sphere2(), measure=_sphere3_measure, trace=True, **kw)
    assert isinstance(got["measure"], mci.Measure) and "obs_add(2," in got["measure"].body
    with pytest.raises(Stop):
        mci.integrate(mci.catalog.sphere2(), measure=_sphere3_measure, trace=False, **kw)
    assert isinstance(got["measure"], mci.HostMeasure)
    with pytest.raises(Stop):   # a measure that adds nothing is a no-op body, not the empty body that means "the default measure"
        mci.integrate(mci.catalog.sphere2(), measure=lambda x, obs, w, c: None, **kw)
    assert isinstance(got["measure"], mci.Measure) and got["measure"].body == "(void)0;"


def test_traced_measures_compile_for_gfx950():
    from mcintegration_jl_amd.trace import trace_measure
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], obs=[0.0, [0.0, 0.0]])
    for m, indexed, solver in ((_sphere3_measure, False, "vegas"), (_sphere3_measure5, True, "mcmc"), (_sphere3_measure, False, "vegasmc"),
                               (_sphere3_measure5, True, "vegas"), (_sphere3_measure5, True, "vegasmc")):
        eng = mci.Engine(cfg, mci.catalog.sphere2(), measure=trace_measure(m, cfg, indexed=indexed), device=-1)
        eng.compile(solver)
        assert os.path.exists(eng.code_object(solver))
        eng.close()
    cfg2 = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3)), dof=[[1, 1]], obs=[[0.0, 0.0]])
    for solver in ("vegas", "vegasmc", "mcmc"):               # a masked measure: comparisons, `!`, selects
        eng = mci.Engine(cfg2, mci.Integrand("return x[0] * x[1];"), measure=trace_measure(_binned, cfg2), device=-1)
        eng.compile(solver)
        assert os.path.exists(eng.code_object(solver))
        eng.close()


def _random_expression(rng, depth):
    """a random closure over three draws, built from closures (not source text): what trace_integrand sees is ordinary Python / numpy"""
    if depth == 0 or rng.random() < 0.15:
        k = int(rng.integers(0, 4))
        if k == 3:
            v = float(rng.choice([0.5, 2.0, -1.25, 3.0, 1e-3]))
            return lambda x: v
        return lambda x, k=k: x[k]
    a, b = _random_expression(rng, depth - 1), _random_expression(rng, depth - 1)
    op = int(rng.integers(0, 12))
    return [lambda x: a(x) + b(x), lambda x: a(x) - b(x), lambda x: a(x) * b(x), lambda x: a(x) / (1.5 + np.abs(b(x))),
            lambda x: np.exp(-np.abs(a(x))), lambda x: np.sin(a(x)) * np.cos(b(x)), lambda x: np.sqrt(np.abs(a(x)) + 1.0),
            lambda x: np.log(np.abs(a(x)) + 1.0), lambda x: where(a(x) > b(x), a(x), b(x) * 0.5), lambda x: fmax(a(x), 0.25) - fmin(b(x), 0.75),
            lambda x: (a(x) + 0.0) ** 2 + np.tanh(b(x)), lambda x: -a(x) + 1.0 * b(x)][op]


@pytest.mark.parametrize("seed", range(12))
def test_random_closures_trace_to_what_they_compute(oracle, seed):
    rng = np.random.default_rng(4000 + seed)
    config = mci.Configuration(var=mci.Continuous(-1.0, 2.0), dof=[[3], [3]])
    e0, e1 = _random_expression(rng, 5), _random_expression(rng, 4)
    f = lambda x, c: (e0(x), e1(x))
    I = trace_integrand(f, config)
    fn = _c_function(oracle, I)
    X = rng.uniform(-1.0, 2.0, size=(3, 100))
    for p in range(100):
        x = np.ascontiguousarray(X[:, p])
        w = np.zeros(2)
        fn(x.ctypes.data_as(C.POINTER(C.c_double)), w.ctypes.data_as(C.POINTER(C.c_double)), None)
        ref = np.array([float(e0(x)), float(e1(x))])
        np.testing.assert_allclose(w, ref, rtol=1e-12, atol=1e-300, err_msg="seed %d at %s\n%s" % (seed, x, I.body))


def _random_branching_closure(rng, ndraw):
    """a closure made of real Python control flow on its draws: nested ternaries, `and` / `or` / `not` with their short circuits, early
    returns; at most six truth tests on any way through it"""
    budget = [6]

    def value(depth):
        kind = rng.integers(0, 6 if depth < 3 and budget[0] > 0 else 3)
        if kind == 0:
            i = int(rng.integers(0, ndraw))
            return lambda x: x[i]
        if kind == 1:
            v = float(rng.uniform(-2.0, 2.0))
            return lambda x: v
        if kind == 2:
            a, b = value(depth + 1), value(depth + 1)
            op = int(rng.integers(0, 3))
            return (lambda x: a(x) + b(x)) if op == 0 else (lambda x: a(x) * b(x)) if op == 1 else (lambda x: a(x) - 0.5 * b(x))
        budget[0] -= 1
        c, a, b = cond(depth + 1), value(depth + 1), value(depth + 1)
        if kind == 3:
            return lambda x: a(x) if c(x) else b(x)
        if kind == 4:
            def early(x):
                if c(x):
                    return a(x)
                return b(x) * 2.0
            return early
        return lambda x: (a(x) if c(x) else b(x)) + (1.0 if c(x) else 0.0)      # the same test met twice on one way

    def cond(depth):
        kind = rng.integers(0, 4 if depth < 3 and budget[0] > 1 else 1)
        if kind == 0:
            a, b = value(depth + 1), value(depth + 1)
            return lambda x: a(x) < b(x)
        budget[0] -= 1
        c1, c2 = cond(depth + 1), cond(depth + 1)
        return (lambda x: c1(x) and c2(x)) if kind == 1 else (lambda x: c1(x) or c2(x)) if kind == 2 else (lambda x: not c1(x))
    body = value(0)
    return lambda x, c: body(x)


def test_random_closures_with_python_control_flow_trace_to_what_they_compute(oracle):
    """trace.explore on 40 random closures built from real Python control flow (nested ternaries, early returns, and / or / not with
    their short circuits, the same test met twice): the written-out body, compiled with gcc, against the closure on plain floats"""
    rng = np.random.default_rng(2026)
    dp = C.POINTER(C.c_double)
    config = mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[4]])
    traced = selects = 0
    for case in range(40):
        f = _random_branching_closure(rng, 4)
        try:
            I = trace_integrand(f, config)
        except TraceError as e:
            assert "ways through" in str(e), (case, e)      # (only the bound on the number of ways may refuse one of these)
            continue
        traced += 1
        selects += I.body.count("?")
        fn = _c_function(oracle, I)
        for _ in range(60):
            x = rng.uniform(-1.0, 1.0, 4)
            w = np.zeros(1)
            fn(x.ctypes.data_as(dp), w.ctypes.data_as(dp), I.userdata.ctypes.data_as(dp) if len(I.userdata) else None)
            assert w[0] == pytest.approx(float(f(x, config)), rel=1e-13, abs=1e-300), (case, x, I.body)
    assert traced >= 35 and selects >= 40, (traced, selects)


def test_ways_through_a_branch_share_what_they_have_in_common(oracle):
    """trace._join: `c ? f(u, k) : f(v, k)` is written `f(c ? u : v, k)` -- the reference's Green's function (test/bubble.jl:40-51), four ways
    with two exponentials each, comes out with three exponentials of selected arguments (the denominator is common to the signs of tau)
    where joining the ways at the result would evaluate eight; same numbers as the closure to the last bit of the division."""
    beta = 6.5

    def green(tau, omega, beta):
        if tau >= 0.0:
            return np.exp(-omega * tau) / (1 + np.exp(-omega * beta)) if omega > 0.0 else np.exp(omega * (beta - tau)) / (1 + np.exp(omega * beta))
        return -np.exp(-omega * (tau + beta)) / (1 + np.exp(-omega * beta)) if omega > 0.0 else -np.exp(-omega * tau) / (1 + np.exp(omega * beta))
    f = lambda x, c: green(x[0], x[1], beta) * x[2] * 3.0
    cfg = mci.Configuration(var=mci.Continuous(-2.0, 2.0), dof=[[3]])
    I = trace_integrand(f, cfg)
    assert I.body.count("exp(") == 3 and I.body.count(" / ") == 1, I.body
    assert I.body.count("* x[2]") == 1                                   # the common factors once, behind the selects
    fn = _c_function(oracle, I)
    rng = np.random.default_rng(8)
    ud = np.ascontiguousarray(I.userdata, dtype=np.float64)
    for _ in range(300):
        x = rng.uniform(-2.0, 2.0, 3)
        w = np.zeros(1)
        fn(x.ctypes.data_as(C.POINTER(C.c_double)), w.ctypes.data_as(C.POINTER(C.c_double)), ud.ctypes.data_as(C.POINTER(C.c_double)))
        assert w[0] == pytest.approx(float(f(x, cfg)), rel=2e-15)
