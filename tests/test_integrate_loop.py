"""The per-iteration loop of integrate() (the path every communicator other than the single-process default takes; main.jl:142-207) on a
scripted engine, without a GPU: the warm-up of the automatic :mcmc chain length -- an iteration whose chains were too short for the holds
they measured is discarded and run again on the Philox streams of iteration + 16384 * attempt, until the first launch that is long enough;
the first iteration of a call that ignores it anyway is let through; nothing is repeated once a launch has been valid -- and what the
Result is built from (block means -> lineage error when the iterations continued each other's chains)."""
import numpy as np

import mcintegration_jl_amd as mci
from mcintegration_jl_amd._lib import MCMC, VEGAS
from mcintegration_jl_amd.comm import LocalComm


class OneRank(LocalComm):   # a communicator type integrate() does not special-case: the per-iteration loop
    pass


class ScriptedEngine:
    """records what the loop asks for; `valid_from` = the attempt (counted over the whole call) from which launches are long enough"""
    nobs = 1

    def __init__(self, config, integrand, valid_from=0, **kw):
        self.calls, self.discards, self.launch, self.valid_from, self.warm = [], 0, -1, valid_from, False
        self.rows = []

    def set_reweight_goal(self, goal):
        pass

    def reset_block_log(self):
        self.rows = []

    def run(self, solver, nevalperblock, lo, hi, iteration, seed, measurefreq, nchain, thermal_ratio):
        self.launch += 1
        self.calls.append((solver, iteration))

    def finish(self, solver, block, adapt, gamma):
        self.rows.append(np.full((4, 1), 1.0 + 0.01 * self.launch))
        return np.array([1.0 + 0.01 * self.launch]), np.array([0.1])

    def last_chain_launch(self):
        return 64, True

    def mcmc_launch_valid(self):
        valid = self.launch >= self.valid_from
        self.warm = self.warm or valid
        return valid, self.warm, 4096, 256

    def discard_iteration(self):
        self.discards += 1
        self.rows.pop()

    def block_means(self, rows):
        return np.array(self.rows[-rows:]), rows - 1


def _run(valid_from, **kw):
    box = {}

    def factory(config, integrand, **k):
        box["eng"] = ScriptedEngine(config, integrand, valid_from=valid_from)
        return box["eng"]
    args = dict(var=mci.Continuous(0.0, 1.0), dof=[[1]], solver="mcmc", neval=4000, niter=5, block=4, comm=OneRank(), engine_factory=factory, print=-1)
    args.update(kw)
    res = mci.integrate("return x[0];", **args)
    return res, box["eng"]


def test_warm_up_iterations_are_run_again_until_one_is_long_enough():
    res, eng = _run(valid_from=3)
    # iteration 0 is ignored by the call (ignore = 1) and let through; iteration 1 takes launches 1, 2 (too short) and 3 (valid); then 2, 3, 4
    assert [it for _, it in eng.calls] == [0, 1, 1 + 16384, 1 + 2 * 16384, 2, 3, 4]
    assert eng.discards == 2 and res.warmup == 2 and res.iter_mean.shape == (5, 1)
    np.testing.assert_allclose(res.iter_mean[:, 0], [1.0, 1.03, 1.04, 1.05, 1.06])       # the counted launches: 0, 3, 4, 5, 6
    assert res.correlated and res.block_mean.shape == (5, 4, 1)


def test_a_call_that_counts_its_first_iteration_repeats_that_one_too():
    res, eng = _run(valid_from=2, ignore=0)
    assert [it for _, it in eng.calls] == [0, 16384, 2 * 16384, 1, 2, 3, 4] and res.warmup == 2


def test_nothing_is_repeated_under_an_explicit_chain_count_or_another_solver():
    res, eng = _run(valid_from=99, nchain=8)
    assert [it for _, it in eng.calls] == [0, 1, 2, 3, 4] and res.warmup == 0
    res, eng = _run(valid_from=99, solver="vegasmc")
    assert [it for _, it in eng.calls] == [0, 1, 2, 3, 4] and res.warmup == 0


def test_a_launch_that_never_gets_long_enough_gives_up_after_seven_repeats():
    res, eng = _run(valid_from=10**9, niter=3)
    its = [it for _, it in eng.calls]
    assert its[:9] == [0] + [1 + 16384 * k for k in range(8)]          # iteration 1: the launch and seven repeats
    assert res.warmup == 14 and res.iter_mean.shape == (3, 1)          # ... iteration 2 likewise: the engine is still not warm


def test_solver_seam_runs_one_block_and_leaves_its_sums_on_the_configuration(oracle):
    """Vegas / VegasMC / MCMC .montecarlo(config, integrand, neval, ...) (src/main.jl:253-264): one block; `_block!` then reads
    observable, normalization, neval and visited off the Configuration (:269-287).  Oracle-backed engine: the numbers are the oracle's
    packed block, the next call is the next stream."""
    from oracle_engine import OracleEngine
    for solver, ns in (("vegas", mci.Vegas), ("vegasmc", mci.VegasMC), ("mcmc", mci.MCMC)):
        cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], seed=5)
        out = ns.montecarlo(cfg, mci.catalog.sphere2(), 4000, 0, [], False, measurefreq=1, engine_factory=OracleEngine, nchain=1)
        assert out is cfg and cfg.iterations_done == 1 and (cfg.neval == 4000 if solver == "vegas" else 0 < cfg.neval <= 4000)   # (chain solvers count like the reference: what the packed block says)
        eng = cfg._engine
        assert eng.calls == [(0, 1, 0)]
        pk = eng.get_packed()
        assert cfg.normalization == pk[2 * eng.nobs] and cfg.normalization > 0.0
        m = np.array(cfg.observable) / cfg.normalization                               # main.jl:275-287
        np.testing.assert_allclose(m, pk[:2], rtol=1e-14)
        assert abs(m[0] - np.pi / 4) < 0.1 and abs(m[1] - np.pi / 6) < 0.1
        assert cfg.visited.shape == (3,) and (solver == "vegas" or cfg.visited.sum() > 1000)
        first = np.array(cfg.observable)
        ns.montecarlo(cfg, mci.catalog.sphere2(), 4000, engine_factory=OracleEngine, nchain=1)
        assert eng is cfg._engine and eng.calls[-1] == (0, 1, 1) and not np.array_equal(first, cfg.observable)   # the same engine, the next stream
    import pytest
    with pytest.raises(ValueError):
        mci.MCMC.montecarlo(mci.Configuration(), mci.catalog.log_over_sqrt(), 100, inplace=True, engine_factory=OracleEngine)
