"""Closures that index a table with a sampled Discrete value and measures that bin by one, the way the reference's histogram examples
are written: docs/src/index.md "Measure Histogram" (`r = grid[bin[1]]`, `obs[1][bin[1]] += weights[1]`) and test/bubble.jl:53-92
(`q = para.extQ[extidx]`, `obs[1][Ext[1]] += weight[1]`).  CPU side: written-out bodies against the closures through gcc, the host
trampolines' integer Discrete draws and accumulating observables.  The GPU side is tests/test_hip_reference_examples.py."""
import ctypes as C
import types

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd import TraceError, trace_integrand
from mcintegration_jl_amd.trace import trace_measure

from test_callback_forms import _engine_like
from test_trace import _c_function

dp = C.POINTER(C.c_double)
N = 20
GRID = [i / N for i in range(1, N + 1)]                       # docs: grid = [i / N for i in 1:N]


def histogram_integrand(vars, config):                        # docs/src/index.md "Measure Histogram" (0-based: bin[0] - 1)
    grid = config.userdata
    x, bin = vars
    r = grid[bin[0] - 1]
    r1 = x[0] ** 2 + r ** 2 < 1
    r2 = x[0] ** 2 + x[1] ** 2 + r ** 2 < 1
    return r1, r2


def histogram_measure(vars, obs, weights, config):
    x, bin = vars
    obs[0][bin[0] - 1] += weights[0]
    obs[1][bin[0] - 1] += weights[1]


def histogram_config(userdata=GRID):
    return mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, N)), dof=[[1, 1], [2, 1]], obs=[np.zeros(N), np.zeros(N)], userdata=userdata)


def _check_body(oracle, I, f, cfg, lo, hi, discrete, n=200, seed=0):
    fn = _c_function(oracle, I)
    rng = np.random.default_rng(seed)
    ud = np.ascontiguousarray(I.userdata, dtype=np.float64)
    for _ in range(n):
        x = rng.uniform(lo, hi)
        for k, (a, b) in discrete.items():
            x[k] = float(rng.integers(a, b + 1))
        w = np.zeros(cfg.N)
        fn(np.ascontiguousarray(x).ctypes.data_as(dp), w.ctypes.data_as(dp), ud.ctypes.data_as(dp) if len(ud) else None)
        yield x, w


def test_a_table_in_userdata_indexed_with_a_discrete_draw(oracle):
    cfg = histogram_config()
    I = trace_integrand(histogram_integrand, cfg)
    assert "ud[0 + (int)fmin(fmax(" in I.body and ", 0.0), 19.0)]" in I.body and list(I.userdata) == GRID
    for x, w in _check_body(oracle, I, histogram_integrand, cfg, np.zeros(3), np.ones(3), {2: (1, N)}):
        ref = histogram_integrand((x[:2], np.array([int(x[2])])), cfg)
        assert w[0] == float(ref[0]) and w[1] == float(ref[1])
    # a numpy array and a tuple as userdata trace to the same body
    assert trace_integrand(histogram_integrand, histogram_config(np.array(GRID))).body == I.body
    assert trace_integrand(histogram_integrand, histogram_config(tuple(GRID))).body == I.body
    # under :mcmc's form
    I3 = trace_integrand(lambda idx, v, c: histogram_integrand(v, c)[idx], cfg, indexed=True)
    assert I3.body.count("ud[0 + (int)") >= 1


def test_rows_of_a_struct_of_parameters(oracle):
    """test/bubble.jl:53-68: `para = config.userdata`, `q = para.extQ[extidx]` a momentum VECTOR, `kq = k + q`"""
    para = types.SimpleNamespace(kF=1.2, me=0.5, extQ=[np.array([0.1 * i, -0.05 * i, 0.0]) for i in range(5)])

    def f(vars, config):
        K, Ext = vars
        p = config.userdata
        k = np.array([K[0], K[1], K[2]])
        q = p.extQ[Ext[0] - 1]
        kq = k + q
        return (np.dot(kq, kq) - p.kF ** 2) / (2 * p.me) + q[1]
    cfg = mci.Configuration(var=(mci.Continuous(-1.0, 1.0), mci.Discrete(1, 5)), dof=[[3, 1]], userdata=para)
    I = trace_integrand(f, cfg)
    # the struct's floats are parameters (kF^2 and 2 me: one ud slot each), the table of momenta follows them: 2 + 5 x 3 values
    assert "3 * (int)fmin(fmax(" in I.body and len(I.userdata) == 17 and list(I.userdata[:2]) == [1.2 ** 2, 1.0]
    for x, w in _check_body(oracle, I, f, cfg, -np.ones(4), np.ones(4), {3: (1, 5)}):
        assert w[0] == pytest.approx(f((x[:3], np.array([int(x[3])])), cfg), rel=1e-13)
    para2 = types.SimpleNamespace(kF=0.7, me=0.25, extQ=[np.array([0.3 * i, 0.0, 0.1]) for i in range(5)])          # another point of a sweep: the same body
    I2 = trace_integrand(f, mci.Configuration(var=(mci.Continuous(-1.0, 1.0), mci.Discrete(1, 5)), dof=[[3, 1]], userdata=para2))
    assert I2.body == I.body and list(I2.userdata[:2]) == [0.7 ** 2, 0.5]
    # a dict of parameters, a two-dimensional array indexed [draw, column]
    dcfg = mci.Configuration(var=(mci.Continuous(-1.0, 1.0), mci.Discrete(1, 5)), dof=[[1, 1]], userdata={"extQ": np.array(para.extQ), "s": 2.0})
    g = lambda v, c: c.userdata["extQ"][v[1][0] - 1, 1] * v[0][0] * c.userdata["s"]
    Ig = trace_integrand(g, dcfg)
    for x, w in _check_body(oracle, Ig, g, dcfg, -np.ones(2), np.ones(2), {1: (1, 5)}):
        assert w[0] == pytest.approx(g((x[:1], np.array([int(x[1])])), dcfg), rel=1e-13)


def test_captured_tables_small_and_large(oracle):
    """an array the closure captured: up to 64 elements they are parameters (ud slots, one body for every value) AND a table when a
    draw indexes them; larger ones are tables only"""
    small, large = np.array([1.0, 2.5, -3.0]), np.linspace(0.0, 1.0, 100) ** 2
    cfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3)), dof=[[1, 1]])
    f = lambda v, c: small[v[1][0] - 1] * v[0][0] + small[0]
    I = trace_integrand(f, cfg)
    assert list(I.userdata) == [1.0, 1.0, 2.5, -3.0] and "ud[1 + (int)" in I.body and "+ ud[0]" in I.body
    for x, w in _check_body(oracle, I, f, cfg, np.zeros(2), np.ones(2), {1: (1, 3)}):
        assert w[0] == pytest.approx(f((x[:1], np.array([int(x[1])])), cfg), rel=1e-14)
    small[:] = [4.0, 5.0, 6.0]                                                     # another value: the same body
    I2 = trace_integrand(f, cfg)
    assert I2.body == I.body and list(I2.userdata) == [4.0, 4.0, 5.0, 6.0]
    cfg2 = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(0, 99)), dof=[[1, 1]])
    g = lambda v, c: large[v[1][0]] * v[0][0]
    Ig = trace_integrand(g, cfg2)
    assert len(Ig.userdata) == 100 and "ud[0 + (int)fmin(fmax(x[1], 0.0), 99.0)]" in Ig.body
    for x, w in _check_body(oracle, Ig, g, cfg2, np.zeros(2), np.ones(2), {1: (0, 99)}):
        assert w[0] == pytest.approx(g((x[:1], np.array([int(x[1])])), cfg2), rel=1e-14)


def test_a_table_over_two_discrete_draws(oracle):
    """`vertex[a, b]` with a sampled value on several leading axes: the row-major flat index of the clamped entries, one lookup"""
    V = np.arange(12.0).reshape(3, 4) ** 1.5
    W = np.arange(24.0).reshape(2, 3, 4) - 7.0
    cfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3), mci.Discrete(1, 4)), dof=[[1, 1, 1]])
    f = lambda v, c: V[v[1][0] - 1, v[2][0] - 1] * v[0][0] + W[1, v[1][0] - 1, v[2][0] - 1] + W[0, v[1][0] - 1][2] + V[v[1][0] - 1][3]
    I = trace_integrand(f, cfg)
    assert len(I.userdata) == 12 + 24                                                # each table once, however it is indexed
    for x, w in _check_body(oracle, I, f, cfg, np.zeros(3), np.ones(3), {1: (1, 3), 2: (1, 4)}):
        ref = f((x[:1], np.array([int(x[1])]), np.array([int(x[2])])), cfg)
        assert w[0] == pytest.approx(ref, rel=1e-14)


def test_what_a_sampled_index_cannot_do():
    cfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3)), dof=[[1, 1]])
    plain = [1.0, 2.0, 3.0]
    with pytest.raises(TraceError, match="indexed with a sampled value"):          # a Python list the tracer cannot see into
        trace_integrand(lambda v, c: plain[v[1][0] - 1], cfg)
    tab = np.arange(12.0).reshape(3, 4)
    with pytest.raises(TraceError, match="behind a slice"):                         # a draw behind a slice
        trace_integrand(lambda v, c: tab[:, v[1][0]][0], cfg)
    with pytest.raises(TraceError, match="disagree"):                               # Python's negative index: the body clamps, the closure wraps
        trace_integrand(lambda v, c: tab[v[1][0] - 3, 0], cfg)


def test_measures_that_bin_by_a_discrete_draw():
    cfg = histogram_config()
    body = trace_measure(histogram_measure, cfg).body
    lines = [ln.strip() for ln in body.splitlines()]
    assert "if (mci_k0_0 >= 0 && mci_k0_0 < 20) obs_add(0 + mci_k0_0, rw[0]);" in lines
    assert "if (mci_k0_1 >= 0 && mci_k0_1 < 20) obs_add(20 + mci_k0_1, rw[1]);" in lines
    # :mcmc's five-argument form (test/bubble.jl:89-92), 0-based idx
    b5 = trace_measure(lambda idx, v, obs, w, c: obs[idx].__setitem__(v[1][0] - 1, obs[idx][v[1][0] - 1] + w), cfg, indexed=True).body
    assert "if (idx < 0 || idx == 1) {" in b5 and "obs_add(20 + mci_k1_0, rw[1]);" in b5
    # a fixed slot and a sampled one of the same observable, complex weights
    ccfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 4)), dof=[[1, 1]], obs=[np.zeros(4, dtype=complex)], type=complex)

    def mc(v, obs, w, c):
        obs[0][v[1][0] - 1] += w[0] * v[0][0]
        obs[0][0] += w[0]
    bc = trace_measure(mc, ccfg).body
    assert "obs_add(0, rw[0]);" in bc and "obs_add(0 + 2 * mci_k0_0 + 1," in bc
    # a branch on a draw around the add: joined with a select
    def mb(v, obs, w, c):
        if v[0][0] < 0.5:
            obs[0][v[1][0] - 1] += w[0]
        else:
            obs[0][v[1][0] - 1] += 2 * w[0]
    assert "? rw[0] :" in trace_measure(mb, cfg).body
    with pytest.raises(TraceError, match="some ways|on another"):            # (a sampled bin on one way through a branch only)
        def half(v, obs, w, c):
            if v[0][0] < 0.5:
                obs[0][v[1][0] - 1] += w[0]
        trace_measure(half, cfg)


def test_host_closures_get_integer_discrete_draws_and_accumulating_observables():
    """trace=False: a Discrete pool reaches a host closure as integers (the reference's Discrete holds Ints), so `grid[bin[0] - 1]` and
    `obs[0][bin[0] - 1] += w` run as written -- over a batch of records too, repeated bins included"""
    from mcintegration_jl_amd.engine import Engine
    cfg = histogram_config(np.array(GRID))
    n = 50
    rng = np.random.default_rng(2)
    X = np.ascontiguousarray(np.vstack([rng.uniform(0, 1, (2, n)), rng.integers(1, N + 1, (1, n)).astype(float)]))
    x, b = _engine_like(cfg)._pool_views(X, n)
    assert x.dtype == np.float64 and b.dtype == np.int64 and b.shape == (1, n) and np.array_equal(b[0], X[2].astype(int))
    W = np.zeros((2, n))
    cb = Engine._make_host_callback(_engine_like(cfg), histogram_integrand, False)
    assert cb(X.ctypes.data_as(dp), W.ctypes.data_as(dp), n, 3, 2, None) == 0
    r = np.array(GRID)[X[2].astype(int) - 1]
    assert np.array_equal(W[0], (X[0] ** 2 + r ** 2 < 1) * 1.0) and np.array_equal(W[1], (X[0] ** 2 + X[1] ** 2 + r ** 2 < 1) * 1.0)
    # the measure over a block's records
    R = np.ascontiguousarray(rng.standard_normal((2, n)))
    O = np.zeros(2 * N)
    mcb = Engine._make_host_measure_callback(_engine_like(cfg), histogram_measure)
    assert mcb(X.ctypes.data_as(dp), R.ctypes.data_as(dp), n, n, 3, 2, 0, O.ctypes.data_as(dp), 2 * N, None) == 0
    ref = np.zeros(2 * N)
    for j in range(n):
        ref[int(X[2, j]) - 1] += R[0, j]
        ref[N + int(X[2, j]) - 1] += R[1, j]
    np.testing.assert_allclose(O, ref, rtol=1e-13, atol=1e-15)
    # a measure written per record with a scalar slot (`obs[1][1] += weights[1]`) raises on arrays and is called record by record
    cfg1 = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, N)), dof=[[1, 1], [2, 1]], obs=[0.0, np.zeros(2)])

    def per_record(v, obs, w, c):
        obs[0][0] += w[0]
        obs[1][1 if v[1][0] > 10 else 0] += w[1]
    O = np.zeros(3)
    mcb = Engine._make_host_measure_callback(_engine_like(cfg1), per_record)
    with pytest.warns(RuntimeWarning, match="called record by record: correct, but slow"):       # (said once)
        assert mcb(X.ctypes.data_as(dp), R.ctypes.data_as(dp), n, n, 3, 2, 0, O.ctypes.data_as(dp), 3, None) == 0
    np.testing.assert_allclose(O, [R[0].sum(), R[1][X[2] <= 10].sum(), R[1][X[2] > 10].sum()], rtol=1e-13)


def test_random_closures_with_tables_inside_branches(oracle):
    """table lookups by a Discrete draw inside nested Python branches (the select is pushed down into the index where two ways look
    the same table up, trace._join): 30 random closures over two Continuous and two Discrete draws, written-out body through gcc against
    the closure on plain numbers"""
    rng = np.random.default_rng(77)
    small, large = rng.uniform(-1.0, 1.0, 6), rng.uniform(-1.0, 1.0, 90)
    rows = rng.uniform(-1.0, 1.0, (6, 2))
    cfg = mci.Configuration(var=(mci.Continuous(-1.0, 1.0), mci.Discrete(1, 6)), dof=[[2, 2]])

    def make():
        budget = [5]

        def value(depth):
            kind = rng.integers(0, 8 if depth < 3 and budget[0] > 0 else 5)
            if kind == 0:
                i = int(rng.integers(0, 2))
                return lambda x, d: x[i]
            if kind == 1:
                v = float(rng.uniform(-2.0, 2.0))
                return lambda x, d: v
            if kind == 2:
                j = int(rng.integers(0, 2))
                return (lambda x, d: small[d[j] - 1]) if rng.random() < 0.5 else (lambda x, d: large[d[j] * 15 - 15 + d[1 - j]])
            if kind == 3:
                j, k = int(rng.integers(0, 2)), int(rng.integers(0, 2))
                return lambda x, d: rows[d[j] - 1][k] * x[k]
            if kind == 4:
                a, b = value(depth + 1), value(depth + 1)
                op = int(rng.integers(0, 3))
                return (lambda x, d: a(x, d) + b(x, d)) if op == 0 else (lambda x, d: a(x, d) * b(x, d)) if op == 1 else (lambda x, d: np.exp(a(x, d)) - 0.5 * b(x, d))
            budget[0] -= 1
            c, a, b = cond(depth + 1), value(depth + 1), value(depth + 1)
            if kind == 5:
                return lambda x, d: a(x, d) if c(x, d) else b(x, d)
            if kind == 6:
                j = int(rng.integers(0, 2))
                return lambda x, d: small[d[j] - 1] * a(x, d) if c(x, d) else small[d[1 - j] - 1] * a(x, d)      # the same table, the index differs by the way
            return lambda x, d: np.exp(a(x, d)) if c(x, d) else np.exp(b(x, d))

        def cond(depth):
            kind = rng.integers(0, 3)
            if kind == 0:
                a, b = value(depth + 1), value(depth + 1)
                return lambda x, d: a(x, d) < b(x, d)
            if kind == 1:
                j, v = int(rng.integers(0, 2)), int(rng.integers(1, 7))
                return lambda x, d: d[j] == v
            j = int(rng.integers(0, 2))
            return lambda x, d: d[j] > d[1 - j]
        body = value(0)
        return lambda v, c: body(v[0], v[1])
    traced = lookups = 0
    for case in range(30):
        f = make()
        try:
            I = trace_integrand(f, cfg)
        except TraceError as e:
            assert "ways through" in str(e), (case, e)
            continue
        traced += 1
        lookups += I.body.count("(int)fmin(")
        fn = _c_function(oracle, I)
        ud = np.ascontiguousarray(I.userdata, dtype=np.float64)
        for _ in range(40):
            x = np.concatenate([rng.uniform(-1.0, 1.0, 2), rng.integers(1, 7, 2).astype(float)])
            w = np.zeros(1)
            fn(x.ctypes.data_as(dp), w.ctypes.data_as(dp), ud.ctypes.data_as(dp) if len(ud) else None)
            ref = float(f((x[:2], x[2:].astype(np.int64)), cfg))
            assert w[0] == pytest.approx(ref, rel=1e-13, abs=1e-300), (case, x, I.body)
    assert traced >= 25 and lookups >= 20, (traced, lookups)


def test_observables_keep_their_axes():
    """an N-d observable (`obs = [zeros(3, 4)]`: a histogram over two Discrete draws) is indexed by the measure as it was declared --
    traced (row-major flat bin, -1 where an entry is off its axis), on the host (batch and per record) and in Result"""
    from mcintegration_jl_amd.engine import Engine
    from mcintegration_jl_amd.statistics import Result
    cfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 3), mci.Discrete(1, 4)), dof=[[1, 1, 1]], obs=[np.zeros((3, 4))])
    assert cfg.obs_shape == [(3, 4)] and cfg.obs_len == [12]

    def m(v, obs, w, c):
        obs[0][v[1][0] - 1, v[2][0] - 1] += w[0]
        obs[0][0, 1] += w[0] * v[0][0]
    body = trace_measure(m, cfg).body
    assert "obs_add(1, " in body and "if (mci_k0_0 >= 0 && mci_k0_0 < 12) obs_add(0 + mci_k0_0, rw[0]);" in body and "? (-1.0) :" in body
    with pytest.raises(TraceError, match="one index per axis"):
        trace_measure(lambda v, obs, w, c: obs[0].__setitem__(v[1][0] - 1, w[0]), cfg)
    n = 40
    rng = np.random.default_rng(4)
    X = np.ascontiguousarray(np.vstack([rng.uniform(0, 1, (1, n)), rng.integers(1, 4, (1, n)).astype(float), rng.integers(1, 5, (1, n)).astype(float)]))
    R = np.ascontiguousarray(rng.standard_normal((1, n)))
    ref = np.zeros((3, 4))
    for j in range(n):
        ref[int(X[1, j]) - 1, int(X[2, j]) - 1] += R[0, j]
        ref[0, 1] += R[0, j] * X[0, j]
    for closure in (m, lambda v, obs, w, c: (np.add.at(obs[0], (v[1][0] - 1, v[2][0] - 1), w[0]), np.add.at(obs[0], (0, 1), (w[0] * v[0][0]).sum()))):
        O = np.zeros(12)
        cb = Engine._make_host_measure_callback(_engine_like(cfg), closure)
        assert cb(X.ctypes.data_as(dp), R.ctypes.data_as(dp), n, n, 3, 1, 0, O.ctypes.data_as(dp), 12, None) == 0
        np.testing.assert_allclose(O.reshape(3, 4), ref, rtol=1e-13, atol=1e-15)
    res = Result(np.arange(24.0).reshape(2, 12), np.ones((2, 12)), cfg, 0)
    assert np.asarray(res.mean[0]).shape == (3, 4) and np.asarray(res.iterations[0][0][0]).shape == (3, 4)
