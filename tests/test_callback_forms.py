"""The callback forms of the drop-in boundary: integrate() calls a closure the way the reference's solver calls it --
`integrand(var, config)`, `integrand(var, weights, config)` with `inplace = true`, `integrand(idx, var, config)` under `:mcmc`
(src/main.jl:26-28; vegas/montecarlo.jl:140-144, vegas_mc/updates.jl:67-75, mcmc/montecarlo.jl:34-36) -- decided by solver + flag,
with the closure's own parameter count as a cross-check that raises.  On the CPU: the decision table, the errors, the traced in-place
bodies of the reference's own two in-place tests (TestComplex2_inplace test/montecarlo.jl:187-196, TestHyperSphere :204-216) against
the closures through gcc, the host trampoline's writable `weights` view, and the traced hypersphere end to end on the oracle."""
import ctypes as C
import math
import types

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd.integrate import callback_form
from mcintegration_jl_amd.trace import TraceError, trace_integrand


def volume_inverse(d):          # test/montecarlo.jl:205-208
    return (d / (2 * math.pi * math.e)) ** (d / 2) * math.sqrt(d) * math.sqrt(math.pi)


def hypersphere_inplace(x, w, c):          # test/montecarlo.jl:210-216, 0-based
    _w = x[0] ** 2
    for i in range(c.userdata):
        _w = _w + x[i + 1] ** 2
        w[i] = np.where(_w < 1.0, volume_inverse(i + 2), 0.0)


def complex2_inplace(x, f, c):             # test/montecarlo.jl:188-192
    f[0] = x[0]
    f[1] = x[0] ** 2 * 1j


def hypersphere_config(N=3):
    return mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[i + 2] for i in range(N)], userdata=N)


class Stop(Exception):
    pass


def _capture():
    got = {}

    def factory(config, integrand, measure=None, **kw):
        got["integrand"], got["measure"] = integrand, measure
        raise Stop()
    return got, factory


def test_the_form_follows_solver_and_flag_like_the_reference():
    """main.jl:26-28: :mcmc -> integrand(idx, var, config); otherwise inplace ? integrand(var, weights, config) : integrand(var, config)"""
    f2, f3 = (lambda x, c: 0.0), (lambda a, b, c: 0.0)
    for solver in ("vegas", "vegasmc"):
        assert callback_form(f2, solver, False) == "plain"
        assert callback_form(f3, solver, True) == "inplace"
    assert callback_form(f3, "mcmc", False) == "indexed"
    assert callback_form(f3, "mcmc", True) == "indexed"          # "inplace: only useful for the :vegas and :vegasmc solver" (main.jl:41)
    m4, m5 = (lambda x, o, w, c: None), (lambda i, x, o, w, c: None)
    assert callback_form(m4, "vegas", what="measure") == callback_form(m4, "vegasmc", what="measure") == "plain"
    assert callback_form(m5, "mcmc", what="measure") == "indexed"
    assert callback_form(m4, "vegas", True, what="measure") == "plain"      # (inplace is about the integrand)
    # defaulted / keyword-only / variadic parameters are the closure's own business
    assert callback_form(lambda x, c, scale=2.0: 0.0, "vegas") == "plain"
    assert callback_form(lambda x, c, scale=2.0: 0.0, "vegas", True) == "inplace"      # (callable with three)
    assert callback_form(lambda *a: 0.0, "mcmc") == "indexed"
    assert callback_form(lambda x, c, *, tag=None: 0.0, "vegasmc") == "plain"
    assert callback_form(print, "vegas") == "plain"                                   # no signature: trusted
    # the engine's extension: a form forced under any solver
    assert callback_form(f3, "vegas", form="indexed") == "indexed"
    assert callback_form(f2, "mcmc", form="plain") == "plain"
    assert callback_form(m5, "vegas", form="indexed", what="measure") == "indexed"
    with pytest.raises(ValueError):
        callback_form(f2, "vegas", form="tuple")


@pytest.mark.parametrize("solver,inplace,fn,says", [
    ("vegas", False, lambda x, w, c: None, "integrand(var, config)"),             # the verdict's case: was silently read as integrand(idx, var, config)
    ("vegasmc", False, lambda x, w, c: None, "integrand(var, config)"),
    ("vegas", True, lambda x, c: 0.0, "integrand(var, weights, config)"),
    ("vegasmc", True, lambda x, c: 0.0, "integrand(var, weights, config)"),
    ("mcmc", False, lambda x, c: 0.0, "integrand(idx, var, config)"),
    ("mcmc", True, lambda x, c: 0.0, "integrand(idx, var, config)"),
    ("vegas", False, lambda: 0.0, "integrand(var, config)"),
])
def test_a_closure_that_contradicts_solver_and_flag_raises(solver, inplace, fn, says):
    """never a silent fall to another form (and never to the host path): TypeError before anything is built, naming the call the
    reference's solver makes and the keywords that choose it"""
    got, factory = _capture()
    with pytest.raises(TypeError) as e:
        mci.integrate(fn, var=mci.Continuous(0.0, 1.0), dof=[[1]], solver=solver, inplace=inplace, engine_factory=factory, print=-1)
    msg = str(e.value)
    assert says in msg and "inplace" in msg and "integrand_form" in msg and "main.jl:26-28" in msg
    assert not got                                                   # no engine was asked for
    for trace in (True, False):                                      # the same under both closure paths
        with pytest.raises(TypeError):
            mci.integrate(fn, var=mci.Continuous(0.0, 1.0), dof=[[1]], solver=solver, inplace=inplace, engine_factory=factory, print=-1, trace=trace)


def test_a_measure_that_contradicts_the_solver_raises():
    got, factory = _capture()
    kw = dict(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], engine_factory=factory, print=-1)
    with pytest.raises(TypeError, match=r"measure\(var, obs, relative_weights, config\)"):
        mci.integrate(mci.catalog.sphere2(), measure=lambda i, x, o, w, c: None, solver="vegas", **kw)
    with pytest.raises(TypeError, match=r"measure\(idx, var, obs, relative_weight, config\)"):
        mci.integrate(mci.catalog.sphere2(), measure=lambda x, o, w, c: None, solver="mcmc", **kw)
    with pytest.raises(Stop):                                        # forced: the five-argument form under :vegas
        mci.integrate(mci.catalog.sphere2(), measure=lambda i, x, o, w, c: None, solver="vegas", measure_form="indexed", **kw)
    assert isinstance(got["measure"], mci.Measure)


def test_integrate_hands_the_engine_the_inplace_closure_in_its_form():
    """inplace=True: traced -> an Integrand whose body stores what the closure stored; trace=False -> HostIntegrand(inplace=True);
    a three-parameter closure under :mcmc is still the indexed form"""
    got, factory = _capture()
    kw = dict(var=mci.Continuous(-1.0, 1.0), dof=[[2], [3], [4]], userdata=3, engine_factory=factory, print=-1)
    for solver in ("vegas", "vegasmc"):
        with pytest.raises(Stop):
            mci.integrate(hypersphere_inplace, solver=solver, inplace=True, **kw)
        I = got["integrand"]
        assert isinstance(I, mci.Integrand) and all("w[%d] = " % i in I.body for i in range(3)) and I.body.count("?") == 3
        with pytest.raises(Stop):
            mci.integrate(hypersphere_inplace, solver=solver, inplace=True, trace=False, **kw)
        H = got["integrand"]
        assert isinstance(H, mci.HostIntegrand) and H.inplace and not H.indexed
    with pytest.raises(Stop):
        mci.integrate(lambda idx, x, c: x[0] * (idx + 1.0), solver="mcmc", trace=False, **kw)
    assert got["integrand"].indexed and not got["integrand"].inplace
    with pytest.raises(Stop):                                        # inplace is ignored by :mcmc, like the reference (main.jl:41)
        mci.integrate(lambda idx, x, c: x[0] * (idx + 1.0), solver="mcmc", inplace=True, trace=False, **kw)
    assert got["integrand"].indexed and not got["integrand"].inplace
    with pytest.raises(ValueError):
        mci.HostIntegrand(lambda a, b, c: 0.0, indexed=True, inplace=True)


from test_trace import _c_function      # (an Integrand or body text through gcc; the Integrand's userdata is passed unless another is given)


def test_traced_inplace_bodies_compute_what_the_reference_closures_store(oracle):
    """TestComplex2_inplace (ComplexF64: every weight is its (re, im) pair of slots) and TestHyperSphere (a loop over c.userdata with a
    select) written out by the tracer, compiled with gcc, against the closures on plain numbers"""
    dp = C.POINTER(C.c_double)
    cfg = mci.Configuration(dof=[[1], [1]], type=complex)
    I = trace_integrand(complex2_inplace, cfg, inplace=True)
    assert [ln.strip() for ln in I.body.splitlines()][-4:] == ["w[0] = x[0];", "w[1] = 0.0;", "w[2] = 0.0;", "w[3] = t1;"]
    fn = _c_function(oracle, I)
    rng = np.random.default_rng(3)
    for _ in range(50):
        x = rng.uniform(0.0, 1.0, 1)
        w = np.zeros(4)
        fn(x.ctypes.data_as(dp), w.ctypes.data_as(dp), None)
        z = np.zeros(2, dtype=complex)
        complex2_inplace(x, z, cfg)
        np.testing.assert_allclose(w, [z[0].real, z[0].imag, z[1].real, z[1].imag], rtol=1e-15)
    cfg = hypersphere_config(3)
    I = trace_integrand(hypersphere_inplace, cfg, inplace=True)
    fn = _c_function(oracle, I)
    inside = np.zeros(3)
    for _ in range(400):
        x = rng.uniform(-1.0, 1.0, 4) * 0.75
        w = np.zeros(3)
        fn(x.ctypes.data_as(dp), w.ctypes.data_as(dp), None)
        ref = np.zeros(3)
        hypersphere_inplace(x, ref, cfg)
        np.testing.assert_array_equal(w, ref)
        inside += ref > 0
    assert (inside > 20).all() and (inside < 400).all()               # both branches of every select were seen
    # the body is the device-source twin of the catalog's hand-written one at the same points
    cat = _c_function(oracle, mci.catalog.hypersphere(3).body)
    ud = np.array([3.0])
    for _ in range(100):
        x = rng.uniform(-1.0, 1.0, 4) * 0.75
        a, b = np.zeros(3), np.zeros(3)
        fn(x.ctypes.data_as(dp), a.ctypes.data_as(dp), None)
        cat(x.ctypes.data_as(dp), b.ctypes.data_as(dp), ud.ctypes.data_as(dp))
        np.testing.assert_allclose(a, b, rtol=1e-14)


def test_complex_weights_trace_in_every_form(oracle):
    """Configuration(type=complex): real, complex and mixed values; the plain and the :mcmc form as well as the in-place one"""
    cfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3)), dof=[[1, 1], [1, 0]], type=complex)
    forms = {
        "plain": (lambda v, c: (v[0][0] * v[1][0] + 1j * v[0][0], np.exp(1j * v[0][0]) / (2.0 + 1j) - (1 + 2j) * v[0][0] ** 2), {}),
        "indexed": (lambda i, v, c: v[0][0] * v[1][0] + 1j * v[0][0] if i == 0 else np.exp(1j * v[0][0]) / (2.0 + 1j) - (1 + 2j) * v[0][0] ** 2, dict(indexed=True)),
    }

    def inpl(v, w, c):
        w[0] = v[0][0] * v[1][0] + 1j * v[0][0]
        w[1] = np.exp(1j * v[0][0]) / (2.0 + 1j) - (1 + 2j) * v[0][0] ** 2
    forms["inplace"] = (inpl, dict(inplace=True))
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(8)
    bodies = set()
    for name, (f, kw) in forms.items():
        I = trace_integrand(f, cfg, **kw)
        bodies.add(I.body)
        fn = _c_function(oracle, I)
        for _ in range(50):
            x = np.array([rng.uniform(0.0, 1.0), float(rng.integers(1, 4))])
            w = np.zeros(4)
            fn(x.ctypes.data_as(dp), w.ctypes.data_as(dp), None)
            z0 = x[0] * x[1] + 1j * x[0]
            z1 = np.exp(1j * x[0]) / (2.0 + 1j) - (1 + 2j) * x[0] ** 2
            np.testing.assert_allclose(w, [z0.real, z0.imag, z1.real, z1.imag], rtol=1e-13, err_msg=name)
    assert len(bodies) == 1                                           # three forms of one function: one body, one code object
    with pytest.raises(TraceError):                                   # a complex weight where the configuration's weights are real
        trace_integrand(lambda x, c: x[0] * 1j, mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]))
    with pytest.raises(TraceError):                                   # complex numbers are not ordered
        trace_integrand(lambda x, c: mci.trace.where(x[0] * 1j > 0.5, 1.0, 0.0), mci.Configuration(dof=[[1]], type=complex))


def test_closures_address_a_pool_with_offset_like_the_reference():
    """Continuous(0, 1, size; offset = 2): the reference's closures read X[i + offset] (variable.jl:577; Sphere2 / Sphere3 at offset = 2,
    test/montecarlo.jl:19-32, :270, :309, :347).  The closure's view of such a pool has `offset` leading slots nobody samples (zeros);
    the kernels' flat draws stay 0-based (x[0] is the first SAMPLED slot)."""
    off = 2
    cfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0, 2 + off, offset=off),), dof=[[2], [3]], neighbor=[(1, 3), (1, 2)])
    f = lambda X, c: (1.0 if X[0 + off] ** 2 + X[1 + off] ** 2 < 1.0 else 0.0, X[2 + off] + X[0])
    body = trace_integrand(f, cfg).body
    assert "x[0] * x[0]" in body and "x[1] * x[1]" in body and "w[1] = x[2];" in body      # X[0] is an unsampled slot: 0
    ns = _engine_like(cfg)
    n = 5
    X = np.ascontiguousarray(np.random.default_rng(3).uniform(0.0, 1.0, (3, n)))
    v = ns._pool_views(X, n)
    assert v.shape == (3 + off, n) and np.all(v[:off] == 0.0) and np.array_equal(v[off:], X)
    s = ns._pool_views(X[:, 1:2], 1, scalar=True)
    assert s.shape == (3 + off,) and np.array_equal(s[off:], X[:, 1])
    # several pools, each with its own offset: a CompositeVar (indexed [leaf][slot] like the reference's, offset 1) and a Discrete (offset 2)
    C = mci.CompositeVar(mci.Continuous(0.0, 1.0), mci.Continuous(0.0, 2.0), offset=1, size=6)
    cfg2 = mci.Configuration(var=(C, mci.Discrete(1, 4, 5, offset=2)), dof=[[2, 1]])

    def f2(V, c):
        (x, y), d = V                                    # `x, y = cvar` (variable.jl:436-447)
        return x[0 + 1] * y[1 + 1] + d[0 + 2]            # leaf x of the first sampled slot, leaf y of the second; the Discrete's first draw
    body = trace_integrand(f2, cfg2).body
    assert "x[0] * x[3]" in body and "+ x[4]" in body
    cv, d = _engine_like(cfg2)._pool_views(np.arange(5.0)[:, None] + np.zeros((5, n)), n)
    assert cv.shape == (2, 3, n) and np.all(cv[:, 0] == 0.0) and d.shape[0] == 3 and np.all(d[:2] == 0.0) and np.all(d[2] == 4.0)
    assert np.all(cv[0][1] == 0.0) and np.all(cv[1][2] == 3.0)


def test_complex_measure_closures_trace_to_re_im_slots():
    """measure(var, obs, relative_weights, config) with ComplexF64 weights and observables (main.jl:279,284): every weight is rw[2 i] +
    i rw[2 i + 1], every observable entry two obs_add slots; the written-out body against the closure at random records (trace_measure's
    own check) and its slots spelled out"""
    from mcintegration_jl_amd.trace import trace_measure
    cfg = mci.Configuration(dof=[[1], [1]], type=complex, obs=[0j, [0j, 0j]])

    def m(x, obs, w, c):
        obs[0][0] += w[0].sum()
        obs[1][0] += (w[1] * 1j).sum()
        obs[1][1] += (w[1] * x[0]).sum()
    body = trace_measure(m, cfg).body
    lines = [ln.strip() for ln in body.splitlines()]
    assert "obs_add(0, rw[0]);" in lines and "obs_add(1, rw[1]);" in lines and "obs_add(3, rw[2]);" in lines     # (w[1] i).re = -w[1].im, .im = w[1].re
    assert any(ln.startswith("const double t") and ln.endswith("= -rw[3];") for ln in lines) and "rw[2] * x[0]" in body and "rw[3] * x[0]" in body
    assert body.count("obs_add(") == 6
    m5 = trace_measure(lambda i, x, obs, w, c: obs[i].__setitem__(0, obs[i][0] + w.conjugate()), cfg, indexed=True).body
    assert "if (idx < 0 || idx == 0) {" in m5 and "if (idx < 0 || idx == 1) {" in m5 and "-rw[" in m5
    with pytest.raises(TraceError):                                    # a complex observable where the configuration's are real
        trace_measure(lambda x, obs, w, c: obs[0].__setitem__(0, w[0] * 1j), mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]))


def test_inplace_closures_the_tracer_refuses():
    cfg = hypersphere_config(3)
    with pytest.raises(TraceError):                                   # a store past the vector
        trace_integrand(lambda x, w, c: w.__setitem__(3, x[0]), cfg, inplace=True)
    # a Python branch on a draw is written out as a select (trace.explore): the reference's own `w[i] = _w < 1.0 ? v : 0.0`
    I = trace_integrand(lambda x, w, c: w.__setitem__(0, 1.0 if x[0] > 0 else 0.0), cfg, inplace=True)
    assert "? 1.0 : 0.0" in I.body
    # an entry the closure never stores is a zero weight
    I = trace_integrand(lambda x, w, c: w.__setitem__(1, x[0]), cfg, inplace=True)
    assert "w[0] = 0.0;" in I.body and "w[1] = x[0];" in I.body and "w[2] = 0.0;" in I.body
    # slices and read-modify-write
    def f(x, w, c):
        w[:] = x[0]
        w[2] += x[1]
        w[0:2] = [x[1], x[2]]
    I = trace_integrand(f, cfg, inplace=True)
    assert "w[0] = x[1];" in I.body and "w[1] = x[2];" in I.body and "x[0] + x[1]" in I.body


def _engine_like(cfg):
    """what Engine's trampoline builders need of an engine, without a device: the configuration and the pool views"""
    from mcintegration_jl_amd.engine import Engine
    ns = types.SimpleNamespace(config=cfg)
    ns._pool_views = types.MethodType(Engine._pool_views, ns)
    return ns


def test_host_trampoline_hands_the_closure_a_writable_weights_view():
    """trace=False: mci_set_integrand_host's callback receives the library's output array; in the in-place form the closure writes
    into a view of it (real weights: no copy), complex weights through a complex scratch array split into the (re, im) rows"""
    from mcintegration_jl_amd.engine import Engine
    dp = C.POINTER(C.c_double)
    n = 7
    rng = np.random.default_rng(1)
    cfg = hypersphere_config(3)
    seen = {}

    def spy(x, w, c):
        seen["w"] = w
        hypersphere_inplace(x, w, c)
    cb = Engine._make_host_callback(_engine_like(cfg), spy, True)
    X = np.ascontiguousarray(rng.uniform(-0.8, 0.8, (4, n)))
    W = np.full((3, n), 123.0)
    assert cb(X.ctypes.data_as(dp), W.ctypes.data_as(dp), n, 4, 3, None) == 0
    assert seen["w"].shape == (3, n) and np.shares_memory(seen["w"], W)
    r2 = np.cumsum(X ** 2, axis=0)[1:]
    np.testing.assert_array_equal(W, np.where(r2 < 1.0, np.array([volume_inverse(d) for d in (2, 3, 4)])[:, None], 0.0))
    # an entry the closure does not store is zero, not what the buffer held
    cb = Engine._make_host_callback(_engine_like(cfg), lambda x, w, c: w.__setitem__(1, x[0]), True)
    W[:] = 9.0
    assert cb(X.ctypes.data_as(dp), W.ctypes.data_as(dp), n, 4, 3, None) == 0
    np.testing.assert_array_equal(W, [np.zeros(n), X[0], np.zeros(n)])
    # complex
    ccfg = mci.Configuration(dof=[[1], [1]], type=complex)
    cb = Engine._make_host_callback(_engine_like(ccfg), complex2_inplace, True)
    X = np.ascontiguousarray(rng.uniform(0.0, 1.0, (1, n)))
    W = np.full((4, n), 5.0)
    assert cb(X.ctypes.data_as(dp), W.ctypes.data_as(dp), n, 1, 4, None) == 0
    np.testing.assert_array_equal(W, [X[0], np.zeros(n), np.zeros(n), X[0] ** 2])
    # an exception in the closure never unwinds through the C frame: status 1
    cb = Engine._make_host_callback(_engine_like(cfg), lambda x, w, c: 1 / 0, True)
    assert cb(X.ctypes.data_as(dp), W.ctypes.data_as(dp), n, 4, 3, None) == 1


def hypersphere_ternary(x, w, c):          # test/montecarlo.jl:210-216 word for word: `w[i] = _w < 1.0 ? volume_inverse(i + 1) : 0.0`
    _w = x[0] ** 2
    for i in range(c.userdata):
        _w += x[i + 1] ** 2
        w[i] = volume_inverse(i + 2) if _w < 1.0 else 0.0


def test_closures_with_python_branches_run_on_both_paths(oracle):
    """The reference's closures branch on their draws with plain ternaries (Sphere1-3, TestHyperSphere).  Traced: every way through the
    branches is run once and the ways are joined with selects (trace.explore) -- the same body as the np.where spelling.  Host path:
    numpy refuses the truth value of a batch, so the trampoline calls such a closure sample by sample."""
    from mcintegration_jl_amd.engine import Engine
    cfg = hypersphere_config(3)
    Ia, Ib = trace_integrand(hypersphere_ternary, cfg, inplace=True), trace_integrand(hypersphere_inplace, cfg, inplace=True)
    a, b = Ia.body, Ib.body
    assert a.count("?") == b.count("?") == 3                                # three selects either way ...
    fa, fb = _c_function(oracle, Ia), _c_function(oracle, Ib)
    dpp = C.POINTER(C.c_double)
    prng = np.random.default_rng(4)
    for _ in range(200):                                                    # ... computing the same weights
        x = prng.uniform(-1.0, 1.0, 4) * 0.8
        wa, wb = np.zeros(3), np.zeros(3)
        fa(x.ctypes.data_as(dpp), wa.ctypes.data_as(dpp), None)
        fb(x.ctypes.data_as(dpp), wb.ctypes.data_as(dpp), None)
        np.testing.assert_array_equal(wa, wb)
    sphere = lambda x, c: 1.0 if x[0] ** 2 + x[1] ** 2 < 1.0 else 0.0       # test/montecarlo.jl:19-23
    assert "? 1.0 : 0.0" in trace_integrand(sphere, mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])).body
    # host trampoline: the batch call raises "truth value of an array ...": sample by sample from then on
    dp = C.POINTER(C.c_double)
    n = 9
    rng = np.random.default_rng(2)
    calls = []

    def counted(x, w, c):
        calls.append(np.ndim(x[0]))
        hypersphere_ternary(x, w, c)
    cb = Engine._make_host_callback(_engine_like(cfg), counted, True)
    X = np.ascontiguousarray(rng.uniform(-0.8, 0.8, (4, n)))
    W = np.full((3, n), 7.0)
    import warnings
    for trip in range(2):
        with warnings.catch_warnings(record=True) as said:                  # the switch to per-sample calls is announced, once
            warnings.simplefilter("always")
            assert cb(X.ctypes.data_as(dp), W.ctypes.data_as(dp), n, 4, 3, None) == 0
        assert len(said) == (1 if trip == 0 else 0) and (trip or "called sample by sample: correct, but slow" in str(said[0].message))
        r2 = np.cumsum(X ** 2, axis=0)[1:]
        np.testing.assert_array_equal(W, np.where(r2 < 1.0, np.array([volume_inverse(d) for d in (2, 3, 4)])[:, None], 0.0))
    assert calls == [1] + [0] * (2 * n)                                     # one refused batch call, then scalars only
    cfg1 = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    cb = Engine._make_host_callback(_engine_like(cfg1), sphere, False)
    X = np.ascontiguousarray(rng.uniform(0.0, 1.0, (2, n)))
    W = np.zeros((1, n))
    with pytest.warns(RuntimeWarning, match="sample by sample"):
        assert cb(X.ctypes.data_as(dp), W.ctypes.data_as(dp), n, 2, 1, None) == 0
    np.testing.assert_array_equal(W[0], (X[0] ** 2 + X[1] ** 2 < 1.0) * 1.0)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc"])
def test_traced_inplace_hypersphere_end_to_end_on_the_oracle(oracle, solver):
    """TestHyperSphere(neval, alg, 3) (test/montecarlo.jl:204-216, run at :333 and :383) as an in-place CLOSURE: integrate() traces it,
    the oracle integrates the written-out body, the result is inside the reference's 7 sigma of its known answers"""
    from oracle_engine import OracleEngine

    class Traced(OracleEngine):
        def __init__(self, config, integrand, **kw):
            assert isinstance(integrand, mci.Integrand)
            integrand.name = oracle.compile_c_integrand(integrand.body)
            super().__init__(config, integrand, **kw)
    r = mci.integrate(hypersphere_inplace, var=mci.Continuous(-1.0, 1.0), dof=[[i + 2] for i in range(3)], userdata=3, neval=100000, print=-1,
                      solver=solver, inplace=True, engine_factory=Traced, seed=18)
    expect = np.array([0.9230, 0.94724, 0.96118])
    got, err = np.array(r.mean, dtype=float).ravel(), np.array(r.stdev, dtype=float).ravel()
    assert (np.abs(got - expect) < 7 * err).all() and (err < 0.02).all(), (got, err)
