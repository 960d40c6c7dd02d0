#!/usr/bin/env python3
"""Development tool (GPU box): randomised-layout parity campaign, wider than tests/test_hip_random_configs.py.

Every case draws a layout (1-5 pools of Continuous / Discrete / CompositeVar, 1-4 integrands with ragged dof tables, grids of 17 to
2000 increments), a launch shape (blocks, steps per block, chains per block, measure cadence, iteration number) and a generator
(Philox4x32-10 or -7), JIT-compiles the three sample-batch kernels for it and compares one iteration of each solver with the oracle on
the same Philox streams: packed sums and histograms to 1e-9 relative, holding-time histogram bucket by bucket, then a three-iteration
:vegas run with train! in between.  Failures are collected, not fatal.   usage: fuzz_layouts.py [--pipe | --persist] [first_case] [ncases]
(--pipe: layouts of the pipelined :vegas loop only; --persist: whole integrate() calls over one Continuous variable type run as ONE
persistent launch, against the oracle's loop)"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import mcintegration_jl_amd as mci
import mci_oracle as oracle

SEED = 20260930


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


from layout_cases import check_persistent_call, pipe_case  # noqa: E402  (shared with tests/test_hip_steady_state.py, test_hip_persistent.py)


PIPE_MODE = False
PERSIST_MODE = False
npersist = 0


def run_persist_case(case_id):
    """(tests/layout_cases.py check_persistent_call, shared with tests/test_hip_persistent.py)"""
    global npersist
    what, persistent = check_persistent_call(oracle, case_id, SEED)
    npersist += 1 if persistent else 0
    return what


def run_case(case_id):
    rng = np.random.default_rng(7000 + case_id)
    var, oleaves, dof, body, ndraw = (pipe_case if PIPE_MODE else random_case)(rng)
    rounds = int(rng.choice([10, 10, 7]))
    nblk = int(rng.integers(1, 6))
    nepb = int(rng.choice([600, 2400, 5000, 12345] if not PIPE_MODE else [1, 2, 255, 513, 2400, 12345, 40001]))
    nchain = int(rng.choice([1, 8, 64]))
    mfreq = int(rng.choice([1, 1, 3]))
    it = int(rng.integers(0, 50))
    what = "case %d: pools=%d ni=%d ndraw=%d rounds=%d blocks=%d nepb=%d nchain=%d measurefreq=%d" % (
        case_id, len(var), len(dof), ndraw, rounds, nblk, nepb, nchain, mfreq)
    oracle.set_rng_rounds(rounds)
    cfg = mci.Configuration(var=var, dof=dof, seed=SEED)
    bits = int(rng.choice([52, 52, 32])) if PIPE_MODE else 52
    eng = mci.Engine(cfg, mci.Integrand(body), rng_rounds=rounds, **({"rng_bits": 32} if bits == 32 else {}))
    assert eng.ndraw == ndraw, what
    fn = oracle.compile_c_integrand(body)
    solvers = (("vegas", oracle.VEGAS),) if PIPE_MODE else (("vegas", oracle.VEGAS), ("vegasmc", oracle.VEGASMC), ("mcmc", oracle.MCMC))
    what += " bits=%d" % bits
    for solver, osolver in solvers:
        ocfg = oracle.Config(oleaves, dof)
        if bits == 32:
            ocfg.set_rng_bits(32)
        nc = nchain if nchain <= nepb else 1
        got = eng.iteration(solver, nepb, 0, nblk, iteration=it, seed=SEED, measurefreq=mfreq, nchain=nc)
        ref = ocfg.iteration(osolver, fn, None, nepb, 0, nblk, it, SEED, measurefreq=mfreq, nchain=nc)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300, err_msg="%s %s dof=%s\n%s" % (what, solver, dof, body))
        if solver == "mcmc":
            np.testing.assert_array_equal(eng.hold_histogram(), ocfg.hold_hist, err_msg="%s dof=%s" % (what, dof))
    ocfg = oracle.Config(oleaves, dof)
    if bits == 32:
        ocfg.set_rng_bits(32)
    r = eng.integrate("vegas", neval=24000, niter=3, block=8, seed=SEED)
    o = ocfg.integrate(oracle.VEGAS, fn, None, neval=24000, niter=3, block=8, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-6, err_msg=what + " (3 iterations of :vegas)")
    eng.close()
    return what


if __name__ == "__main__":
    if "--pipe" in sys.argv:   # only layouts of the pipelined :vegas loop, :vegas only, odd launch sizes, both stream widths
        sys.argv.remove("--pipe")
        PIPE_MODE = True
    if "--persist" in sys.argv:   # whole integrate() calls over one Continuous variable type as one persistent launch
        sys.argv.remove("--persist")
        PERSIST_MODE = True
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    oracle.build()
    bad, t0 = [], time.time()
    for c in range(first, first + n):
        try:
            w = run_persist_case(c) if PERSIST_MODE else run_case(c)
            print("ok   " + w, flush=True)
        except Exception as e:  # collect and go on
            bad.append(c)
            print("FAIL case %d: %s" % (c, "".join(traceback.format_exception_only(type(e), e))[:3000]), flush=True)
    oracle.set_rng_rounds(10)
    print("%d cases, %d failed %s in %.0f s" % (n, len(bad), bad, time.time() - t0) + (" (%d ran as one persistent launch)" % npersist if PERSIST_MODE else ""))
    mci.shutdown()
