"""CPU-side tests of the product library: the C ABI loads and exports every symbol include/mci.h
declares, the host-side statistics of the path agree with the oracle and the golden vectors, the
JIT produces gfx950 code objects without a GPU, and compute entry points refuse to run without one."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from mcintegration_jl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mci.h")).read()
    declared = set(re.findall(r"\b(mci_[a-z_0-9]+)\s*\(", hdr))
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert declared == bound, declared ^ bound
    L = C.CDLL(_lib.library_path())
    for name in declared:
        assert hasattr(L, name), name
    assert mci.lib().mci_version().startswith(b"mci-hip")


def test_no_device_fails_loudly():
    from mcintegration_jl_amd.engine import device_count
    if device_count() > 0:
        pytest.skip("a GPU is visible")
    p = C.c_void_p()
    rc = mci.lib().mci_ctx_create(0, C.byref(p))
    assert rc == 7  # MCI_ERR_NO_DEVICE: no CPU fallback
    assert b"no CPU fallback" in mci.lib().mci_last_error()
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    eng = mci.Engine(cfg, mci.catalog.x2y2(), device=-1)  # offline: compile only
    with pytest.raises(mci.MCIError) as e:
        eng.run("vegas", 100, 0, 1, 0, 1)
    assert e.value.code == 7


def test_offline_jit_and_problem_info():
    cfg = mci.Configuration(var=(mci.Continuous(0.0, 1.0), mci.Discrete(1, 4)), dof=[[2, 1], [1, 0]])
    eng = mci.Engine(cfg, mci.Integrand("w[0] = x[0] * x[1] * x[2]; w[1] = x[0];"), device=-1)
    eng.compile()
    assert eng.ndraw == 3 and eng.nobs == 2 and eng.table_mode == 0
    assert eng.packed_size == 2 * 2 + 2 + 3 + 999 + 4 + 2 * (3 * 3 * 3)   # [stats | histograms | propose | accept] (configuration.jl:185-186)
    np.testing.assert_allclose(eng.grid(0), np.linspace(0, 1, 1000), atol=1e-15)
    d, a = eng.distribution(1)
    np.testing.assert_allclose(d, 0.25)
    np.testing.assert_allclose(a, [0, 0.25, 0.5, 0.75, 1.0])
    big = mci.Engine(mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]]), mci.catalog.genz_product_peak(32), device=-1)
    # 32 grids = 256 KB of edges + 256 KB of histograms: histograms in LDS (two tiles), edges from L2
    assert big.table_mode == 3 and big.ndraw == 32 and big.lds_bytes <= 160 * 1024


def test_bad_integrand_source_reports_compile_error():
    eng = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]), mci.Integrand("w[0] = undefined_symbol(x[0]);"), device=-1)
    with pytest.raises(mci.MCIError) as e:
        eng.compile()
    assert e.value.code == 3 and "undefined_symbol" in str(e.value)


def test_configuration_dof_forms_and_asserts():
    # configuration.jl:134-151 and test/utility.jl:14-15
    X = mci.Continuous(0.0, 1.0)
    assert mci.Configuration(var=X, dof=2).dof == [[2]]
    assert mci.Configuration(var=X, dof=[2, 3]).dof == [[2], [3]]
    assert mci.Configuration(var=(X, mci.Continuous(0, 1)), dof=np.array([[1, 2], [3, 4]])).dof == [[1, 3], [2, 4]]
    assert mci.Configuration(var=X, dof=[(1,)]).dof == [[1]]   # test/interface_tests.jl
    c = mci.Configuration(var=(mci.Continuous(0, 1), mci.Continuous(0, 1), mci.Continuous(0, 1), mci.Continuous(0, 1)),
                          dof=[[1, 2, 3, 5], [3, 1, 2, 7], [2, 4, 1, 2]])
    assert c.maxdof == [3, 4, 3, 7]
    with pytest.raises(AssertionError):
        mci.Configuration(var=X, dof=[[1, 1]])
    with pytest.raises(AssertionError):
        mci.Configuration(var=X, dof=[[1]], reweight=[1.0, -1.0])
    # pool auto-resize (configuration.jl:156-160; test/montecarlo.jl:41)
    T = mci.Continuous(0.0, 1.0, 2)
    assert T.size == 3
    mci.Configuration(var=(T,), dof=[[2], [3]])
    assert T.size == 5


def test_variable_constructors():
    # test/variable.jl:1-20
    d = mci.Discrete([1, 4])
    assert d.lower == 1 and d.upper == 4
    assert mci.Dist.is_variable(mci.Continuous) and mci.Dist.is_variable(mci.Discrete) and not mci.Dist.is_variable(int)
    x = mci.Continuous(0.0, 1.0, 7, grid=[0.0, 0.1, 0.4, 1.0])
    assert x.size == 8 and x.ninc == 4
    cv = mci.Continuous([(0.0, 1.0), (0.0, 2.0)])
    assert isinstance(cv, mci.CompositeVar) and len(cv) == 2 and cv[1].upper == 2.0 and cv[0].ninc == 1000
    c = mci.Configuration(var=(x, d), dof=[[1, 1], [2, 1]])
    c.iterations_done = 7
    c.reset_seed(99)                                                                     # configuration.jl:196-199
    assert c.seed == 99 and c.iterations_done == 0
    assert c.propose.shape == c.accept.shape == (3, 3, 3) and np.all(c.propose == 1e-8) and not c.accept.any()      # configuration.jl:185-186
    # the fields a reference user reads off a variable before anything has run: uniform map, cleared histogram (variable.jl:565)
    assert np.array_equal(x.grid, [0.0, 0.1, 0.4, 1.0]) and np.array_equal(x.histogram, np.full(3, 1e-10))
    assert np.array_equal(d.histogram, np.full(4, 1e-10)) and np.allclose(d.distribution, 0.25) and d.accumulation[-1] == 1.0


def test_host_statistics_match_oracle_and_golden(oracle, golden):
    v = golden["mean_std"]
    m, e = mci.mean_std(v["obs_sum"], v["obs_sq"], v["block"])
    np.testing.assert_allclose(m, v["mean"], rtol=1e-15)
    np.testing.assert_allclose(e, v["std"], rtol=1e-13)
    t = golden["average_docs_table"]
    for mx in range(1, 11):
        got = mci.average(t["iter_mean"], t["iter_std"], init=t["init"], max=mx)
        assert got == pytest.approx(oracle.average(t["iter_mean"], t["iter_std"], init=t["init"], max=mx), rel=1e-14)
        assert got[0] == pytest.approx(t["printed"][mx - 1][0], abs=6e-8)
    assert mci.integrate.__module__  # importable
    from mcintegration_jl_amd.integrate import standardize_block
    for args in ((10000, 16, 1), (10000, 16, 3), (10000, 2, 8)):
        assert standardize_block(*args) == oracle.standardize_block(*args)
    # doReweight! fixed point, test/mpi_test.jl:148-169
    r = np.array(golden["doreweight"]["reweight0"])
    vis = np.ascontiguousarray(golden["doreweight"]["visited"], dtype=np.float64)
    goal = np.ascontiguousarray(golden["doreweight"]["goal"])
    dp = C.POINTER(C.c_double)
    for _ in range(5):
        mci.lib().mci_do_reweight(r.ctypes.data_as(dp), vis.ctypes.data_as(dp), 4, 1.0, goal.ctypes.data_as(dp))
    np.testing.assert_allclose(r, golden["doreweight"]["expect"], rtol=1e-3)
    out = np.zeros(4, dtype=np.int32)
    dof = np.ascontiguousarray(golden["maxdof"]["dof"], dtype=np.int32)
    ip = C.POINTER(C.c_int32)
    mci.lib().mci_maxdof(dof.ctypes.data_as(ip), 3, 4, out.ctypes.data_as(ip))
    assert list(out) == golden["maxdof"]["expect"]


def test_result_and_report(capsys, golden):
    t = golden["average_docs_table"]
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]])
    res = mci.Result(np.array(t["iter_mean"])[:, None], np.array(t["iter_std"])[:, None], cfg, ignore=1)
    assert res.mean[0] == pytest.approx(-3.9979808, abs=1e-7) and res.stdev[0] == pytest.approx(0.0013607691, rel=1e-6)
    assert res.chi2[0] == pytest.approx(1.9269, abs=1e-4) and res.dof == 8
    mci.report(res)
    text = capsys.readouterr().out
    assert "ignore" in text and "-3.99798 ± 0.00136" in text  # statistics.jl:74-96: digits = 2 - floor(log10(err))
    res3 = res.with_ignore(3)
    assert res3.ignore == 3 and res3.mean[0] != res.mean[0]


def test_plain_c_consumer_of_the_abi_builds_and_refuses_to_run_without_a_gpu():
    """examples/mci_demo.c: the boundary is plain C (pointers and sizes); without a device it exits with
    MCI_ERR_NO_DEVICE instead of falling back to the CPU."""
    import subprocess
    from mcintegration_jl_amd.engine import device_count
    demo = os.path.join(ROOT, "examples", "mci_demo")
    assert os.path.exists(demo), "run `python __graft_entry__.py` (build()) first"
    if device_count() > 0:
        pytest.skip("a GPU is visible")
    out = subprocess.run([demo], capture_output=True, text=True)
    assert out.returncode == 7 and "no CPU fallback" in out.stderr


def test_mcmc_auto_chain_length_rule():
    """mci_mcmc_auto_chains: pilot-length chains (4096 steps or 2 burn-in floors) until a launch has been measured; afterwards
    16 x (fresh) / 4 x (carried) the longest holding time of the launch before, at most 2 x the chain length that measured it (a hold
    longer than a quarter of that chain is censored by it; 0 = no cap), never fewer than 8 / 1 burn-in floors; at most 131072 chains
    per GPU, at least one chain."""
    from mcintegration_jl_amd._lib import lib
    L = lib()
    npb, nblocks, nslots, nd, npool = 6250000, 16, 12, 5, 1
    fl = 64 * nslots + 16 * (npool + 1) * nd
    assert fl == 928
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 0, 0, 0) == npb // 4096                     # nothing measured: 4096 > 2 floors
    assert L.mci_mcmc_auto_chains(npb, nblocks, 64, nd, npool, 0, 0, 0) == npb // (2 * (64 * 64 + 16 * 2 * 5))    # ... 2 floors > 4096
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 256, 0, 0) == npb // max(16 * 256, 8 * fl)      # light tails: the floor decides
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 16384, 0, 0) == npb // (16 * 16384)             # heavy tails: the holds decide
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 16384, 0, 1) == npb // (4 * 16384)              # carried chains: 4 x
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 64, 0, 1) == npb // fl                          # ... and one floor
    # censored: the launch before ran 4096-step chains and saw holds up to 8192 (its top bucket's upper edge) -> 2 x 4096, not 8 x 8192
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 8192, 4096, 1) == npb // (2 * 4096)
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 8192, 4096, 0) == npb // (2 * 4096)
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 512, 16384, 1) == npb // (4 * 512)             # holds that fit: the length comes down at once
    assert L.mci_mcmc_auto_chains(npb, nblocks, nslots, nd, npool, 1 << 30, 0, 0) == 1
    assert L.mci_mcmc_auto_chains(10**9, 16, 1, 2, 1, 2, 0, 0) == 131072 // 16                                     # GPU-fill cap


def test_inverse_variance_weights_from_16_blocks_underestimate_the_error_by_a_seventh():
    """A property of the reference's own combination of iterations (statistics.jl:186-220) that every "scatter / reported error" number
    of this repository has to be read against: the weights 1/sigma_i^2 come from each iteration's block scatter, at the default
    block = 16 a chi^2 with 15 degrees of freedom.  For independent Gaussian iterations of equal variance the seed scatter of the
    weighted mean over the reported error is ~sqrt(nu / (nu - 4)) = 1.17 in the limit of many iterations (1.14 for nine), 1.03 at
    block = 64 -- :vegas included; nothing a sampler does to its chains."""
    rng = np.random.default_rng(1)
    for B, lo, hi in ((16, 1.10, 1.19), (64, 1.0, 1.06)):
        nrep, nit = 4000, 9
        bm = rng.normal(size=(nrep, nit, B))
        im, ie = bm.mean(2), bm.std(2, ddof=1) / np.sqrt(B)
        got = np.array([mci.average(im[r], ie[r], init=1, max=nit)[:2] for r in range(nrep)])
        ratio = got[:, 0].std(ddof=1) / np.sqrt((got[:, 1] ** 2).mean())
        assert lo < ratio < hi, (B, ratio)


def test_lineage_sums_are_the_scatter_of_the_blocks_weighted_averages():
    """mci_lineage_sums (the error of a run whose iterations continued each other's chains): per observable the sum and the sum of
    squares over the blocks of every block's weighted average over the iterations, with the weights of `average` (statistics.jl:197,
    :217) -- through _mean_std (main.jl:296-320) they give the same mean as `average` and the lineages' scatter as its error."""
    from mcintegration_jl_amd.statistics import lineage_sums, mean_std
    rng = np.random.default_rng(5)
    niter, nb, nobs = 7, 64, 3
    # AR(1) lineages: consecutive iterations of a block correlate (rho = 0.85), blocks are independent
    bm = np.zeros((niter, nb, nobs))
    bm[0] = rng.normal(size=(nb, nobs))
    for i in range(1, niter):
        bm[i] = 0.85 * bm[i - 1] + np.sqrt(1 - 0.85 ** 2) * rng.normal(size=(nb, nobs))
    bm += 5.0
    im = bm.mean(1)
    ie = bm.std(1, ddof=1) / np.sqrt(nb)
    for init in (1, 2, 4):
        s1, s2 = lineage_sums(bm, ie, init=init, max=niter)
        w = 1.0 / (ie[init - 1:] + 1e-10) ** 2
        w /= w.sum(0)
        mb = (bm[init - 1:] * w[:, None, :]).sum(0)            # [nb][nobs]
        np.testing.assert_allclose(s1, mb.sum(0), rtol=1e-13)
        np.testing.assert_allclose(s2, (mb ** 2).sum(0), rtol=1e-13)
        m, e = mean_std(s1, s2, nb)
        for o in range(nobs):
            assert m[o] == pytest.approx(mci.average(im[:, o], ie[:, o], init=init, max=niter)[0], rel=1e-12)
        np.testing.assert_allclose(e, mb.std(0, ddof=1) / np.sqrt(nb), rtol=1e-10)
    # a Result built from them: the reference's mean and chi2, the lineage error; Result(res, ignore) recomputes both
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1], [1], [1]])
    res = mci.Result(im, ie, cfg, ignore=1, block_mean=bm, correlated=True, block=nb)
    plain = mci.Result(im, ie, cfg, ignore=1)
    assert res.mean == plain.mean and res.chi2 == plain.chi2
    s1, s2 = lineage_sums(bm, ie, init=2, max=niter)
    np.testing.assert_allclose(res.stdev, mean_std(s1, s2, nb)[1], rtol=1e-14)
    assert all(a > b for a, b in zip(res.stdev, plain.stdev))      # positively correlated iterations: the reference's formula is too small
    r3 = res.with_ignore(3)
    s1, s2 = lineage_sums(bm, ie, init=4, max=niter)
    np.testing.assert_allclose(r3.stdev, mean_std(s1, s2, nb)[1], rtol=1e-14)
    # two "ranks" with half of the blocks each: the sums add (what the library's loop does with mci_comm_sum; integrate() gathers the
    # block means of all ranks once instead, so that a Result is plain data)
    tot = []
    for half in (bm[:, :nb // 2], bm[:, nb // 2:]):
        tot.append(np.concatenate(lineage_sums(half, ie, init=2, max=niter)))
    np.testing.assert_allclose(mean_std((tot[0] + tot[1])[:nobs], (tot[0] + tot[1])[nobs:], nb)[1], res.stdev, rtol=1e-12)
    gathered = np.zeros_like(bm)
    for lo, hi in ((0, nb // 2), (nb // 2, nb)):   # the sum integrate() forms over the ranks: zero outside a rank's own blocks
        part = np.zeros_like(bm)
        part[:, lo:hi] = bm[:, lo:hi]
        gathered += part
    split = mci.Result(im, ie, cfg, ignore=1, block_mean=gathered, correlated=True, block=nb)
    np.testing.assert_allclose(split.stdev, res.stdev, rtol=1e-14)
    assert not hasattr(split, "_sum_ranks") and split.with_ignore(3).stdev == r3.stdev   # local: no communicator, no engine


def test_closure_parameter_count_ignores_parameters_with_defaults():
    """integrate() decides a closure's form by solver + `inplace` like the reference (tests/test_callback_forms.py); the closure's
    REQUIRED positional parameters are the cross-check: a defaulted or keyword-only extra parameter is the closure's own business."""
    from mcintegration_jl_amd.integrate import required_positionals as rp
    assert rp(lambda x, config: 0, 2) == 2
    assert rp(lambda x, config, scale=2.0: 0, 2) == 2            # was counted as three: f would have been called as f(idx, x, config)
    assert rp(lambda idx, x, config: 0, 2) == 3
    assert rp(lambda idx, x, config, *, tag=None: 0, 2) == 3
    assert rp(lambda x, obs, weights, config, norm=1.0: 0, 4) == 4
    assert rp(lambda idx, x, obs, weight, config: 0, 4) == 5

    def with_varargs(x, config, *rest, **kw):
        return 0
    assert rp(with_varargs, 2) == 2
    assert rp(print, 2) in (0, 2)                                # builtins without a signature fall back to the plain form


def test_result_says_when_weighted_and_plain_average_of_the_iterations_disagree():
    """statistics.jl:186-220 weights the iterations by 1 / sigma_i^2; when an iteration's error estimate moves with its mean (heavy
    tails) that average is biased.  The engine reproduces the reference's average and says so: Result.weighting_shift, and a note under
    the table of report(result)."""
    import io
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]])
    rng = np.random.default_rng(5)
    # iterations whose error grows with their value: the low ones get the weight
    im = np.array([[1.0], [1.0], [1.0], [1.0], [3.0], [3.0], [3.0]]) + 0.01 * rng.standard_normal((7, 1))
    ie = np.array([[0.05], [0.05], [0.05], [0.05], [1.0], [1.0], [1.0]])
    res = mci.Result(im, ie, cfg, ignore=0)
    assert res.weighting_shift[0] > 2.0
    assert res.plain_mean[0] == pytest.approx(im[:, 0].mean()) and res.plain_stdev[0] == pytest.approx(im[:, 0].std(ddof=1) / np.sqrt(7))
    out = io.StringIO()
    mci.report(res, io=out)
    assert "note: the weighted average lies" in out.getvalue()
    quiet = mci.Result(1.0 + 0.01 * rng.standard_normal((7, 1)), np.full((7, 1), 0.01), cfg, ignore=0)
    assert quiet.weighting_shift[0] < 2.0
    out = io.StringIO()
    mci.report(quiet, io=out)
    assert "note:" not in out.getvalue()


def test_chain_estimator_bias_note_follows_block_length_and_count():
    """statistics.chain_estimator_bias: z = k tau sqrt(B I / N) for ONE chain per block (the reference's chain); report(result) prints
    its note from z = 2 on.  The +5.7 .. +7.3 sigma call of profiles/r05_odd_calls.txt (:mcmc, block = 256, neval = 1e6: 3906 steps per
    block) with the acceptance measured there gives z = 7; the default call does not reach 2; several chains per block: no estimate."""
    import io
    from mcintegration_jl_amd.statistics import chain_estimator_bias as est
    pr, ac = np.full((3, 5, 5), 100.0), np.full((3, 5, 5), 92.0)
    b = est("mcmc", 3906, 1, 256, 9, pr, ac, 12)
    assert 6.5 < b["z"] < 7.5 and 14.0 < b["tau"] < 16.0 and 200 < b["times"] < 300
    assert est("mcmc", 3906, 4, 256, 9, pr, ac, 12) is None and est("vegas", 3906, 1, 256, 9, pr, ac, 12) is None
    assert est("vegasmc", 3906, 1, 256, 9, pr, ac, 12)["z"] == pytest.approx(b["z"] / 5.0)
    assert est("mcmc", 62500, 1, 16, 9, pr, ac, 12)["z"] == pytest.approx(b["z"] / 16.0, rel=1e-3)     # z ~ B at fixed neval: 16 x fewer blocks of 16 x the length
    assert est("mcmc", 625, 1, 16, 9, np.full(4, 100.0), np.full(4, 90.0), 1)["z"] < 1.0                # the default call, README integrand
    assert est("mcmc", 3906, 1, 256, 9, pr, np.zeros_like(pr), 12) is None                              # nothing accepted yet: no estimate
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]])
    res = mci.Result(1.0 + 0.01 * np.arange(5.0)[:, None], np.full((5, 1), 0.01), cfg, ignore=1)
    out = io.StringIO()
    mci.report(res, io=out)
    assert "note: solver" not in out.getvalue() and res.chain_bias_note is None
    res.chain_bias = b
    out = io.StringIO()
    mci.report(res, io=out)
    assert "note: solver = :mcmc ran one chain per block of 3906 steps" in out.getvalue() and "fewer blocks (block = 16 gives 0.4)" in out.getvalue()
    assert res.with_ignore(2).chain_bias is b


def test_exported_train_helpers_match_the_pinned_oracle(oracle):
    """Dist.locate / smooth / rescale (test/utility.jl:1-10; src/distribution/common.jl): the mirror's host versions against the oracle's,
    which are pinned on the reference's vectors (tests/test_oracle_known_answers.py)"""
    grid = [0.0, 0.1, 0.3, 0.5]
    eps = np.finfo(float).eps
    for p, want in ((eps, 1), (0.5 - eps, 3), (grid[0], 1), (0.05, 1), (0.2, 2), (0.31, 3)):      # test/utility.jl:3-9
        assert mci.Dist.locate(grid, p) == want == oracle.locate(grid, p)
    for bad in (-0.1, 0.5, 0.7):
        with pytest.raises(ValueError):
            mci.Dist.locate(grid, bad)
    rng = np.random.default_rng(9)
    for n in (1, 2, 7, 999):
        h = rng.uniform(0.1, 3.0, n) ** 4
        np.testing.assert_allclose(mci.Dist.smooth(h, 6.0), oracle.smooth(h, 6.0), rtol=1e-15)
        for alpha in (0.5, 1.5, 3.0):
            np.testing.assert_allclose(mci.Dist.rescale(h, alpha), oracle.rescale(h, alpha), rtol=1e-14)
    assert mci.Dist.poolsize(mci.Continuous(0.0, 1.0)) == 17                                       # MaxOrder + 1
