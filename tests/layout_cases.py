"""Randomised problem layouts shared by tests/ and tools/fuzz_layouts.py (data generators, not code under test)."""
import mcintegration_jl_amd as mci


def pipe_case(rng):
    """layouts the software-pipelined :vegas loop takes (mci_device.h pipe_eligible): Continuous pools only, 8..16 draws in all, grids small
    enough for LDS pair tables; ragged dof tables (padding probabilities), adapt on/off, 1-4 integrands"""
    while True:
        npool = int(rng.integers(1, 4))
        ni = int(rng.integers(1, 5))
        dof = [[int(rng.integers(0, 9)) for _ in range(npool)] for _ in range(ni)]
        for i in range(ni):
            if sum(dof[i]) == 0:
                dof[i][int(rng.integers(0, npool))] = 1
        maxdof = [max(d[v] for d in dof) for v in range(npool)]
        if 8 <= sum(maxdof) <= 16 and min(maxdof) > 0:
            break
    var, oleaves = [], []
    for v in range(npool):
        lo, hi = float(rng.uniform(-2, 0)), float(rng.uniform(0.5, 3))
        ninc = int(rng.choice([17, 100, 257, 1000]))
        alpha = float(rng.choice([1.0, 2.0, 3.0]))
        adapt = bool(rng.integers(0, 5) > 0)
        var.append(mci.Continuous(lo, hi, alpha=alpha, ninc=ninc, adapt=adapt))
        oleaves.append(dict(kind=0, pool=v, lower=lo, upper=hi, npts=ninc, alpha=alpha, adapt=adapt))
    draws = [(v, s) for v in range(npool) for s in range(maxdof[v])]
    lines = []
    for i in range(ni):
        own = [k for k, (v, s) in enumerate(draws) if s < dof[i][v]]
        coef = rng.uniform(0.2, 1.5, size=len(own))
        arg = " + ".join("%.6f * x[%d]" % (c, k) for c, k in zip(coef, own))
        sign = "-" if rng.integers(0, 4) == 0 else ""
        lines.append("w[%d] = %s(%.3f + 0.5 * cos(%s) + 0.05 * x[%d] * x[%d]);" % (i, sign, 0.4 + 0.3 * i, arg, own[0], own[-1]))
    return tuple(var), oleaves, dof, "\n".join(lines), len(draws)


def persist_case(rng):
    """one Continuous variable type (the persistent :vegas launch's layouts): random bounds, grid size, learning rate; 1-4 integrands
    with their own dof on the shared pool; an integrand that may change sign"""
    lo, hi = float(rng.uniform(-2, 0)), float(rng.uniform(0.5, 3))
    ninc = int(rng.choice([17, 100, 257, 1000, 1000, 1025, 1500]))
    alpha = float(rng.choice([0.5, 1.0, 2.0, 2.0, 3.0]))
    ni = int(rng.integers(1, 5))
    dof = [[int(rng.integers(1, 7))] for _ in range(ni)]
    ndraw = max(d[0] for d in dof)
    lines = []
    for i in range(ni):
        own = list(range(dof[i][0]))
        coef = rng.uniform(0.2, 1.5, size=len(own))
        arg = " + ".join("%.6f * x[%d]" % (c, k) for c, k in zip(coef, own))
        sign = "-" if rng.integers(0, 4) == 0 else ""
        lines.append("w[%d] = %s(%.3f + 0.5 * cos(%s) + 0.05 * x[%d] * x[%d]);" % (i, sign, 0.4 + 0.3 * i, arg, own[0], own[-1]))
    var = (mci.Continuous(lo, hi, alpha=alpha, ninc=ninc),)
    oleaves = [dict(kind=0, pool=0, lower=lo, upper=hi, npts=ninc, alpha=alpha)]
    return var, oleaves, dof, "\n".join(lines), ndraw


def check_persistent_call(oracle, case_id, seed=20260930):
    """a whole integrate() call as ONE persistent launch (mci_set_persistent on) against the oracle's loop: every iteration's mean and
    error, the trained map; odd sizes, 1-40 blocks, with and without adaptation.  Returns (description, ran persistently?)."""
    import numpy as np
    rng = np.random.default_rng(9000 + case_id)
    var, oleaves, dof, body, ndraw = persist_case(rng)
    block = int(rng.choice([1, 2, 5, 16, 16, 40]))
    neval = int(rng.choice([block * 3 + 1, 1000, 10007, 40000, 123457]))
    neval = max(neval, block + 1)
    niter = int(rng.integers(1, 7))
    adapt = bool(rng.integers(0, 5) > 0)
    what = "case %d: ni=%d ndraw=%d ninc=%d alpha=%g block=%d neval=%d niter=%d adapt=%d" % (
        case_id, len(dof), ndraw, oleaves[0]["npts"], oleaves[0]["alpha"], block, neval, niter, adapt)
    oracle.set_rng_rounds(10)
    cfg = mci.Configuration(var=var, dof=dof, seed=seed)
    eng = mci.Engine(cfg, mci.Integrand(body))
    eng.set_persistent("on")
    fn = oracle.compile_c_integrand(body)
    ocfg = oracle.Config(oleaves, dof)
    r = eng.integrate("vegas", neval=neval, niter=niter, block=block, seed=seed, adapt=adapt, ignore=0)
    persistent = eng.last_integrate_persistent()
    what += " persistent=%d" % persistent
    o = ocfg.integrate(oracle.VEGAS, fn, None, neval=neval, niter=niter, block=block, seed=seed, adapt=adapt, ignore=0)
    np.testing.assert_allclose(r["iter_mean"][0], o["iter_mean"][0], rtol=1e-10, atol=1e-300, err_msg=what + " (first iteration)")
    # later iterations: the prefix-scan walk against the oracle's recurrence.  With fewer samples than bins the histogram is mostly its
    # 1e-10 offsets and train! amplifies rounding without bound: those cases are held to a twentieth of the statistical error instead
    sparse = neval * ndraw < 20 * oleaves[0]["npts"]
    if sparse:
        assert np.all(np.abs(r["iter_mean"] - o["iter_mean"]) <= 0.05 * o["iter_std"] + 1e-5 * np.abs(o["iter_mean"])), what
    else:
        np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-5, atol=1e-300, err_msg=what)
        np.testing.assert_allclose(r["iter_std"], o["iter_std"], rtol=1e-3, atol=1e-300, err_msg=what)
    g, og = eng.grid(0), ocfg.grid(0)
    assert g[0] == og[0] and g[-1] == og[-1] and np.all(np.diff(g) > 0), what
    if not sparse:
        np.testing.assert_allclose(g, og, rtol=0, atol=1e-5 * (og[-1] - og[0]), err_msg=what + " (map)")
    eng.close()
    return what, persistent


def random_case(rng):
    npool = int(rng.integers(1, 6))
    ni = int(rng.integers(1, 5))
    var, oleaves, pool_nleaf = [], [], []
    for v in range(npool):
        kind = rng.choice(["cont", "cont", "disc", "comp", "comp2"])
        if kind == "cont":
            lo, hi = float(rng.uniform(-2, 0)), float(rng.uniform(0.5, 3))
            ninc = int(rng.choice([17, 100, 257, 1000, 2000]))
            alpha = float(rng.choice([0.5, 1.0, 2.0, 3.0]))
            adapt = bool(rng.integers(0, 4) > 0)
            var.append(mci.Continuous(lo, hi, alpha=alpha, ninc=ninc, adapt=adapt))
            oleaves.append(dict(kind=0, pool=v, lower=lo, upper=hi, npts=ninc, alpha=alpha, adapt=adapt))
            pool_nleaf.append(1)
        elif kind == "disc":
            lo = int(rng.integers(0, 3))
            hi = lo + int(rng.integers(0, 9))
            adapt = bool(rng.integers(0, 2))
            var.append(mci.Discrete(lo, hi, adapt=adapt))
            oleaves.append(dict(kind=1, pool=v, lower=lo, upper=hi, adapt=adapt))
            pool_nleaf.append(1)
        elif kind == "comp":
            a = (float(rng.uniform(-1, 0)), float(rng.uniform(0.5, 2)))
            b = (int(rng.integers(1, 3)), int(rng.integers(3, 6)))
            var.append(mci.CompositeVar(mci.Continuous(*a), mci.Discrete(*b)))
            oleaves.append(dict(kind=0, pool=v, lower=a[0], upper=a[1]))
            oleaves.append(dict(kind=1, pool=v, lower=b[0], upper=b[1]))
            pool_nleaf.append(2)
        else:
            a = (float(rng.uniform(-1, 0)), float(rng.uniform(0.5, 2)))
            c = (float(rng.uniform(0, 1)), float(rng.uniform(1.5, 4)))
            b = (int(rng.integers(0, 2)), int(rng.integers(2, 5)))
            n2 = int(rng.choice([50, 1000]))
            var.append(mci.CompositeVar(mci.Continuous(*a), mci.Discrete(*b), mci.Continuous(*c, ninc=n2, alpha=1.5)))
            oleaves.append(dict(kind=0, pool=v, lower=a[0], upper=a[1]))
            oleaves.append(dict(kind=1, pool=v, lower=b[0], upper=b[1]))
            oleaves.append(dict(kind=0, pool=v, lower=c[0], upper=c[1], npts=n2, alpha=1.5))
            pool_nleaf.append(3)
    dof = [[int(rng.integers(0, 5)) for _ in range(npool)] for _ in range(ni)]
    for i in range(ni):
        if sum(dof[i]) == 0:
            dof[i][int(rng.integers(0, npool))] = 1
    maxdof = [max(d[v] for d in dof) for v in range(npool)]
    draws = [(v, s, l) for v in range(npool) for s in range(maxdof[v]) for l in range(pool_nleaf[v])]
    lines = []
    for i in range(ni):
        own = [k for k, (v, s, l) in enumerate(draws) if s < dof[i][v]]
        coef = rng.uniform(0.2, 1.5, size=len(own))
        arg = " + ".join("%.6f * x[%d]" % (c, k) for c, k in zip(coef, own))
        sign = "-" if rng.integers(0, 4) == 0 else ""           # some integrands change sign
        lines.append("w[%d] = %s(%.3f + 0.5 * cos(%s) + 0.05 * x[%d] * x[%d]);" % (i, sign, 0.4 + 0.3 * i, arg, own[0], own[-1]))
    return tuple(var), oleaves, dof, "\n".join(lines), len(draws)


def vary_chain_lanes(eng, case_id):
    """FUZZ_LANES=1 (tools/fuzz_layouts.py --lanes): a random group size and speculation tree per case (csrc/mci_spec.h) instead of the
    automatic choice -- the chain is the same chain on any of them.  Returns a note for the case's description."""
    import os
    import numpy as np
    if not os.environ.get("FUZZ_LANES"):
        return ""
    r = np.random.default_rng(777000 + case_id)
    lanes = int(r.choice([1, 2, 4, 8, 16, 32, 64, -1]))
    accept = float(r.choice([0.0, 1e-3, 0.1, 0.3, 0.5, 0.8]))
    limit = int(r.choice([-1, -1, 1, 2, 3]))
    eng.set_chain_speculation(lanes, accept, limit)
    return " lanes=%d tree=(%g, %d)" % (lanes, accept, limit)


def check_carried_iterations(oracle, case_id, seed=20260930):
    """consecutive iterations of both chain solvers on a random layout (1-5 pools of Continuous / Discrete / CompositeVar, 1-4 integrands,
    ragged dof), with doReweight! and train! in between and a chain count that changes from one iteration to the next: carried chains
    (mci_set_chain_carry; :mcmc: the stored chains resampled to the moved reweight factors, k_resample_chains | mcio_resample_chains)
    against the oracle's mirror -- packed sums and histograms, the reweight factors, the holding-time histogram"""
    import numpy as np
    rng = np.random.default_rng(11000 + case_id)
    var, oleaves, dof, body, ndraw = random_case(rng)
    nblk = int(rng.integers(1, 5))
    npb = int(rng.choice([1200, 2400, 4800]))
    counts = [int(rng.choice([2, 5, 8, 16, 40])) for _ in range(4)]
    mfreq = int(rng.choice([1, 1, 3]))
    what = "case %d: pools=%d ni=%d ndraw=%d blocks=%d npb=%d nchain=%s measurefreq=%d" % (case_id, len(var), len(dof), ndraw, nblk, npb, counts, mfreq)
    oracle.set_rng_rounds(10)
    fn = oracle.compile_c_integrand(body)
    ni = len(dof)
    for solver, osolver in (("vegasmc", oracle.VEGASMC), ("mcmc", oracle.MCMC)):
        cfg = mci.Configuration(var=var, dof=dof, seed=seed)
        eng = mci.Engine(cfg, mci.Integrand(body))
        note = vary_chain_lanes(eng, case_id)
        what += note if note not in what else ""
        ocfg = oracle.Config(oleaves, dof)
        n = eng.nobs
        nstat = 2 * n + 2 + ni + 1
        kw = {}
        if solver == "mcmc":
            ocfg.set_thermal_ratio(0.1)
            kw = dict(thermal_ratio=0.1)
        for it, nch in enumerate(counts):
            got = eng.iteration(solver, npb, 0, nblk, iteration=it, seed=seed, measurefreq=mfreq, nchain=nch, **kw)
            ref = ocfg.iteration(osolver, fn, None, npb, 0, nblk, it, seed, measurefreq=mfreq, nchain=nch)
            assert eng.last_chain_launch() == (nch, it > (1 if solver == "vegasmc" else 0)), (what, solver, it, eng.last_chain_launch())   # (:vegasmc: not out of a launch on the untrained map)
            # (every train! in between amplifies the rounding-level difference of the two maps by an order of magnitude or two)
            np.testing.assert_allclose(got[:nstat], ref[:nstat], rtol=1e-8 * 10 ** it, atol=1e-300, err_msg="%s %s iteration %d" % (what, solver, it))
            np.testing.assert_allclose(got[nstat:], ref[nstat:], rtol=1e-7 * 10 ** it, err_msg="%s %s iteration %d (histograms)" % (what, solver, it))
            if solver == "mcmc":
                np.testing.assert_array_equal(eng.hold_histogram(), ocfg.hold_hist, err_msg="%s iteration %d" % (what, it))
            eng.finish(solver, nblk, adapt=True)
            ocfg.set_reweight(oracle.do_reweight(ocfg.reweight, ref[2 * n + 2: 2 * n + 2 + ni + 1]))
            ocfg.train()
            np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=1e-8 * 10 ** it, err_msg=what)
        eng.close()
    return what


def check_walks_agree(case_id, seed=20260930):
    """train!'s serial walk as slots with given decisions ("serial") against the general form of the same recurrence ("serial_general"):
    deterministic runs (bit-reproducible histograms) of a random layout -- one Continuous variable type or several, grids of 17 to
    1500 increments, learning rates 0.5 .. 3, every third case with a narrow peak (bins that yield many points, then an adapting
    grid) -- must give IDENTICAL iterations and grids."""
    import numpy as np
    rng = np.random.default_rng(9000 + case_id)
    var, oleaves, dof, body, ndraw = (persist_case if case_id % 2 == 0 else pipe_case)(rng)
    if case_id % 3 == 0:
        lo, hi = oleaves[0]["lower"], oleaves[0]["upper"]
        c, k = float(rng.uniform(lo, hi)), float(rng.choice([20.0, 200.0, 2000.0])) / (hi - lo)
        body += "\n{ const double t = (x[0] - (%.6f)) * %.6f; const double pk = exp(-t * t);" % (c, k)
        body += "".join(" w[%d] *= pk;" % i for i in range(len(dof))) + " }"
    neval, niter = int(rng.choice([20000, 60000, 200000])), int(rng.integers(4, 9))
    what = "case %d: pools=%d ni=%d ndraw=%d ninc=%s alpha=%s neval=%d niter=%d%s" % (
        case_id, len(var), len(dof), ndraw, [lf["npts"] for lf in oleaves], [lf["alpha"] for lf in oleaves], neval, niter,
        " peak" if case_id % 3 == 0 else "")
    out = []
    for walk in ("serial", "serial_general"):
        cfg = mci.Configuration(var=var, dof=dof, seed=seed)
        eng = mci.Engine(cfg, mci.Integrand(body), deterministic=True)
        eng.set_train_walk(walk)
        r = eng.integrate("vegas", neval=neval, niter=niter, block=16, seed=seed)
        out.append((r["iter_mean"].copy(), r["iter_std"].copy(), [eng.grid(i) for i in range(len(oleaves))], eng.walk_counts()))
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), what
    for g0, g1 in zip(out[0][2], out[1][2]):
        assert np.array_equal(g0, g1), what
        assert np.all(np.diff(g0) > 0), what
    assert out[1][3][0] == 0 and sum(out[0][3]) == out[1][3][1], (what, out[0][3], out[1][3])   # (one walk per adapting leaf and iteration)
    return what + "   walks as slots %d, general %d" % out[0][3]
