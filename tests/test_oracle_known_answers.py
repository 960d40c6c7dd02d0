"""Pins the CPU oracle (oracle/) against every deterministic known answer the reference's tests hold
for the VEGAS path, against the hand-derived golden vectors in tests/golden/golden.json, and against
the printed sample output of the reference (docs/src/index.md:40-49).  CPU only.

Mirrors /root/reference/test/utility.jl, test/statistics.jl, test/mpi_test.jl:148-169.
"""
import math

import numpy as np
import pytest

RTOL = 1e-13  # second implementation differs only by libm ulps (pow/log)


def test_philox_kat(oracle, golden):
    for v in golden["philox4x32_10"]:
        assert oracle.philox(v["ctr"], v["key"]) == v["out"]


def test_philox_7_round_kat_and_stream(oracle, golden):
    """the opt-in cheaper generator (mci_set_rng_rounds(7)): Philox4x32-7 on the Random123 known-answer vectors, and every stream of
    the oracle switching to it (same counters, keys and bit selection)"""
    for v in golden["philox4x32_7"]:
        assert oracle.philox(v["ctr"], v["key"], rounds=7) == v["out"]
    seed, stream, idx = 0x1234567890ABCDEF, 17, (5 << 32) | 9
    ten = [oracle.uniform(seed, stream, idx, k) for k in range(6)]
    oracle.set_rng_rounds(7)
    try:
        for k in range(6):
            o = oracle.philox([idx & 0xFFFFFFFF, idx >> 32, k >> 1, stream], [seed & 0xFFFFFFFF, seed >> 32], rounds=7)
            a, b = o[2 * (k & 1)], o[2 * (k & 1) + 1]
            assert oracle.uniform(seed, stream, idx, k) == (((b << 32) | a) >> 12) * 2.0 ** -52
            o4 = oracle.philox([idx & 0xFFFFFFFF, idx >> 32, k >> 2, stream], [seed & 0xFFFFFFFF, seed >> 32], rounds=7)
            assert oracle.uniform(seed, stream, idx, k, bits=32) == o4[k & 3] * 2.0 ** -32
    finally:
        oracle.set_rng_rounds(10)
    assert [oracle.uniform(seed, stream, idx, k) for k in range(6)] == ten


def test_uniform_stream_contract(oracle):
    # draw k uses words (2*(k&1), 2*(k&1)+1) of Philox(ctr=(idx lo, idx hi, k>>1, stream), key=seed)
    seed, stream, idx = 0x1234567890ABCDEF, 17, (5 << 32) | 9
    for k in range(6):
        o = oracle.philox([idx & 0xFFFFFFFF, idx >> 32, k >> 1, stream], [seed & 0xFFFFFFFF, seed >> 32])
        a, b = o[2 * (k & 1)], o[2 * (k & 1) + 1]
        expect = (((b << 32) | a) >> 12) * 2.0 ** -52
        assert oracle.uniform(seed, stream, idx, k) == expect
        assert 0.0 <= expect < 1.0


def test_locate(oracle, golden):
    # test/utility.jl:2-9
    g = golden["locate"]["grid"]
    for p, expect in golden["locate"]["cases"]:
        assert oracle.locate(g, p) == expect
    assert oracle.locate(g, 0.5) == -1  # reference raises error() (common.jl:10-12)
    assert oracle.locate(g, -1e-3) == -1


def test_maxdof(oracle, golden):
    # test/utility.jl:14-15
    assert list(oracle.maxdof(golden["maxdof"]["dof"])) == golden["maxdof"]["expect"]


def test_smooth_rescale_golden(oracle, golden):
    for v in golden["smooth"].values():
        np.testing.assert_allclose(oracle.smooth(v["inp"]), v["out"], rtol=RTOL)
    for v in golden["rescale"].values():
        np.testing.assert_allclose(oracle.rescale(v["inp"], v["alpha"]), v["out"], rtol=RTOL)


def test_rescale_asserts(oracle):
    with pytest.raises(AssertionError):
        oracle.rescale([1.0, 0.0, 2.0], 2.0)  # common.jl:71
    # length-1 passes through untouched (common.jl:68-70)
    assert oracle.rescale([3.0], 2.0)[0] == 3.0


def test_train_continuous_golden(oracle, golden):
    for v in golden["train_continuous"].values():
        out = oracle.train_continuous(v["grid"], v["hist"], v["alpha"])
        np.testing.assert_allclose(out, v["out"], rtol=RTOL, atol=1e-15)
        assert out[0] == v["grid"][0] and out[-1] == v["grid"][-1]  # variable.jl:217-218,235
        assert np.all(np.diff(out) > 0)


def test_uniform32_stream_contract(oracle, golden):
    """the opt-in 32-bit stream of :vegas (mci_set_rng_bits): draw k = word k & 3 of Philox block k >> 2, its 32 bits the top
    mantissa bits -> exactly word / 2^32; pinned on the Random123 known-answer blocks"""
    for v in golden["philox4x32_10"]:
        ctr, key = v["ctr"], v["key"]
        if ctr[2] >= 2 ** 30:
            continue   # (the block counter word carries k >> 2)
        seed, idx, stream = key[0] | (key[1] << 32), ctr[0] | (ctr[1] << 32), ctr[3]
        for j in range(4):
            u = oracle.uniform(seed, stream, idx, 4 * ctr[2] + j, bits=32)
            assert u == v["out"][j] / 2.0 ** 32 and 0.0 <= u < 1.0
    seed, stream, idx = 0x0123456789ABCDEF, 8 * 3 + 0, 123456789012
    for k in range(9):
        o = oracle.philox([idx & 0xFFFFFFFF, idx >> 32, k >> 2, stream], [seed & 0xFFFFFFFF, seed >> 32])
        assert oracle.uniform(seed, stream, idx, k, bits=32) == o[k & 3] / 2.0 ** 32
    # a Config switched to 32 bits changes the :vegas stream only
    a = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[2]])
    b = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[2]])
    b.set_rng_bits(32)
    pa, pb = a.iteration(oracle.VEGAS, "x2y2", None, 2000, 0, 2, 0, 7), b.iteration(oracle.VEGAS, "x2y2", None, 2000, 0, 2, 0, 7)
    assert pa[0] != pb[0] and abs(pa[0] - pb[0]) < 0.2
    ma, mb = a.iteration(oracle.VEGASMC, "x2y2", None, 2000, 0, 2, 0, 7), b.iteration(oracle.VEGASMC, "x2y2", None, 2000, 0, 2, 0, 7)
    assert (ma == mb).all()


def test_train_continuous_flat_histogram_keeps_uniform_grid(oracle):
    grid = np.linspace(0.0, 1.0, 1000)
    out = oracle.train_continuous(grid, np.full(999, 1e-10), 2.0)
    np.testing.assert_allclose(out, grid, atol=1e-12)


def test_train_asserts(oracle):
    grid = np.linspace(0.0, 1.0, 6)
    with pytest.raises(AssertionError):
        oracle.train_continuous(grid, [1.0, float("nan"), 1, 1, 1], 2.0)  # variable.jl:212
    with pytest.raises(AssertionError):
        oracle.train_continuous(grid, [1.0, 0.0, 1, 1, 1], 2.0)  # variable.jl:213


def test_train_discrete_golden(oracle, golden):
    for v in golden["train_discrete"].values():
        dist, acc = oracle.train_discrete(v["hist"], v["alpha"])
        np.testing.assert_allclose(dist, v["distribution"], rtol=RTOL)
        np.testing.assert_allclose(acc, v["accumulation"], rtol=RTOL)
        assert acc[0] == 0.0 and abs(acc[-1] - 1.0) < 1e-14  # variable.jl:379


def _hand_config(oracle, golden, prob_mode=0):
    # test/utility.jl:31-35 : X, Y Continuous with hand grids, Z = Discrete(1,6; distribution=rand(6))
    h = golden["hand_grids"]
    rng = np.random.default_rng(3)
    leaves = [dict(kind=0, pool=0, lower=0.0, upper=1.0, grid=h["X"]),
              dict(kind=0, pool=1, lower=0.0, upper=1.0, grid=h["Y"]),
              dict(kind=1, pool=2, lower=1, upper=6, distribution=rng.random(6))]
    return oracle.Config(leaves, h["dof"], prob_mode=prob_mode)


def test_map_draw_golden(oracle, golden):
    cfg = _hand_config(oracle, golden)
    for leaf, key in ((0, "X"), (1, "Y")):
        for v in golden["map_draw"][key]:
            prop = cfg.create(leaf, 1, v["y"])
            assert cfg.pool_data(leaf)[0] == pytest.approx(v["x"], rel=1e-15, abs=1e-300)
            assert cfg.pool_gidx(leaf)[0] == v["gidx"]
            assert cfg.pool_prob(leaf)[0] == pytest.approx(v["prob"], rel=1e-15)
            assert prop == pytest.approx(1.0 / v["prob"], rel=1e-15)


def test_shift_equals_create_and_rollback(oracle, golden):
    # sampler.jl:383-384: prob *= dx_old/dx_new is algebraically 1/(N dx_new)
    cfg = _hand_config(oracle, golden)
    rng = np.random.default_rng(0)
    cfg.create(0, 1, 0.7)
    for _ in range(200):
        u = rng.random()
        before = (cfg.pool_data(0)[0], cfg.pool_gidx(0)[0], cfg.pool_prob(0)[0])
        cfg.shift(0, 1, u)
        g = np.array(golden["hand_grids"]["X"])
        iy = cfg.pool_gidx(0)[0]
        assert cfg.pool_prob(0)[0] == pytest.approx(1.0 / (3 * (g[iy] - g[iy - 1])), rel=1e-12)
        if rng.random() < 0.5:
            oracle.lib().mcio_shift_rollback(cfg.p, 0, 1)
            after = (cfg.pool_data(0)[0], cfg.pool_gidx(0)[0], cfg.pool_prob(0)[0])
            assert after == before


def test_probability_invariant(oracle, golden):
    # test/utility.jl:30-55: total_probability ~ probability(i) * padding_probability(i),
    # after initialize! and again after shift!
    cfg = _hand_config(oracle, golden, prob_mode=1)
    c = cfg.c
    rng = np.random.default_rng(5)
    for vi in range(c.npool):
        P = c.leaf[c.pool_leaf0[vi]].P
        for idx in range(1, P - 2 + 1):  # variable.jl:576-580
            cfg.pool_create(vi, idx, [rng.random()])
    tot = cfg.total_probability()
    for i in range(c.Ni):
        assert tot == pytest.approx(cfg.probability(i) * cfg.padding_probability(i), rel=1e-12)
    for vi in range(c.npool):
        for i in range(1, c.maxdof[vi] + 1):
            cfg.pool_shift(vi, i, [rng.random()])
    tot = cfg.total_probability()
    for i in range(c.Ni):
        assert tot == pytest.approx(cfg.probability(i) * cfg.padding_probability(i), rel=1e-12)
    # normalisation integrand (dof = 0): probability 1, padding = total (configuration.jl:153)
    assert cfg.probability(c.Ni) == 1.0
    assert cfg.padding_probability(c.Ni) == pytest.approx(tot, rel=1e-12)


def test_pool_sizes(oracle):
    # test/variable.jl:19-20 pool length size+1; configuration.jl:156-160 auto-resize to maxdof+2
    cfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[2]])
    assert cfg.leaf(0).P == 17
    cfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[16]])
    assert cfg.leaf(0).P == 18
    assert cfg.leaf(0).nbin == 999 and cfg.leaf(0).npts == 1000  # variable.jl:137,147
    g = cfg.grid(0)
    assert g[0] == 0.0 and g[-1] == 1.0


def test_mean_std(oracle, golden):
    # test/statistics.jl:14-46: _mean_std == (mean, std/sqrt(block))
    v = golden["mean_std"]
    m, e = oracle.mean_std(v["obs_sum"], v["obs_sq"], v["block"])
    np.testing.assert_allclose(m, v["mean"], rtol=1e-15)
    np.testing.assert_allclose(e, v["std"], rtol=1e-13)
    for k, series in enumerate(v["series"]):
        s = np.array(series)
        assert m[k] == pytest.approx(s.mean(), rel=1e-14)
        assert e[k] == pytest.approx(s.std(ddof=1) / math.sqrt(len(s)), rel=1e-12)
    m1, e1 = oracle.mean_std([0.3], [0.09], 1)  # block == 1 -> std 0 (main.jl:310-312)
    assert e1[0] == 0.0


def test_average_against_reference_printed_table(oracle, golden):
    # docs/src/index.md:40-49: the reference's own printed per-iteration table; its "wgt average",
    # error and reduced chi2 columns pin average() (statistics.jl:186-220) to the printed digits.
    v = golden["average_docs_table"]
    for mx in range(1, 11):
        mean, err, chi2 = oracle.average(v["iter_mean"], v["iter_std"], init=v["init"], max=mx)
        pm, pe, pc = v["printed"][mx - 1]
        assert mean == pytest.approx(pm, abs=6e-8)
        assert err == pytest.approx(pe, rel=2e-7)
        assert chi2 == pytest.approx(pc, abs=6e-5)
        cm, ce, cc = v["computed"][mx - 1]
        assert (mean, err, chi2) == pytest.approx((cm, ce, cc), rel=1e-13)


def test_do_reweight_fixed_point(oracle, golden):
    # test/mpi_test.jl:148-169
    v = golden["doreweight"]
    r = np.array(v["reweight0"])
    for _ in range(v["n_iterations"]):
        r = oracle.do_reweight(r, v["visited"], v["gamma"], v["goal"])
    np.testing.assert_allclose(r, v["expect"], rtol=v["rtol"])


def test_standardize_block(oracle):
    # main.jl:220-234
    assert oracle.standardize_block(10000, 16, 1) == (625, 16)
    assert oracle.standardize_block(10000, 16, 3) == (666, 15)
    assert oracle.standardize_block(10000, 2, 8) == (1250, 8)


def test_clear_and_add_statistics(oracle):
    # configuration.jl:238-262, variable.jl:565-567; mirrors test/mpi_test.jl:73-109 (sum over workers)
    leaves = [dict(kind=0, pool=0, lower=0.0, upper=1.0, npts=5),
              dict(kind=0, pool=1, lower=0.0, upper=1.0, npts=4),
              dict(kind=1, pool=1, lower=1, upper=3)]
    a = oracle.Config(leaves, [[1, 1]])
    b = oracle.Config(leaves, [[1, 1]])
    a.clear_statistics()
    b.clear_statistics()
    assert np.all(a.hist(0) == 1e-10) and a.c.normalization == 1e-10 and a.c.neval == 0
    assert np.all(a.visited == 1e-8)
    for cfg, scale in ((a, 1.0), (b, 2.0)):
        for i in range(3):
            h = np.ctypeslib.as_array(cfg.leaf(i).hist, shape=(cfg.leaf(i).nbin,))
            h[:] = scale * (1.1 + 0.1 * i)
        cfg.c.normalization = scale
        cfg.c.neval = int(10 * scale)
    oracle.lib().mcio_add_config(a.p, b.p)
    for i in range(3):
        np.testing.assert_allclose(a.hist(i), 3.0 * (1.1 + 0.1 * i))
    assert a.c.normalization == 3.0 and a.c.neval == 30


# ---------------------------------------------------------------------------------------------
# :mcmc primitives (sampler.jl remove!/swap!, configuration.jl _neighbor, mcmc burn-in)
# ---------------------------------------------------------------------------------------------
def test_default_neighbor_graph(oracle):
    # configuration.jl:203-208, 0-based; the last index is the normalisation integrand
    def nb(n):
        return oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[1]] * n).neighbor()
    assert nb(1) == [[1], [0]]                         # Nd = 2: neighbor[1] = [2], neighbor[end] = [1]
    assert nb(2) == [[2, 1], [0], [0]]                 # Nd = 3: [Nd, 2], [Nd-2], [1]
    assert nb(4) == [[4, 1], [0, 2], [1, 3], [2], [0]]


def test_remove_is_the_probability_of_the_slot_and_swap_is_an_involution(oracle):
    cfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0, npts=4, grid=[0.0, 0.1, 0.4, 1.0]),
                         dict(kind=1, pool=0, lower=1, upper=6, distribution=[1, 2, 3, 4, 5, 6])], [[3]])
    for idx, (u0, u1) in enumerate([(0.2, 0.05), (0.5, 0.5), (0.9, 0.99)], start=1):
        inv = cfg.pool_create(0, idx, [u0, u1])        # create! returns 1/prob (sampler.jl:304, :21)
        assert cfg.pool_remove(0, idx) == pytest.approx(1.0 / inv, rel=1e-15)   # remove! returns prob (:322, :39)
    before = (cfg.pool_data(0).copy(), cfg.pool_data(1).copy(), cfg.pool_prob(0).copy(), cfg.pool_gidx(0).copy())
    assert cfg.pool_swap(0, 1, 3) == 1.0
    assert cfg.pool_data(0)[0] == before[0][2] and cfg.pool_data(0)[2] == before[0][0]
    assert cfg.pool_data(1)[0] == before[1][2] and cfg.pool_gidx(0)[0] == before[3][2]
    cfg.pool_swap(0, 1, 3)                              # swapRollback! == swap!  (sampler.jl:403-408)
    after = (cfg.pool_data(0), cfg.pool_data(1), cfg.pool_prob(0), cfg.pool_gidx(0))
    for a, b in zip(before, after):
        assert np.array_equal(a, b)


def test_mcmc_burnin_rule(oracle):
    L = oracle.lib()
    assert L.mcio_mcmc_burnin(62500, 1, 3, 3, 1, 0.1) == 6250          # mcmc/montecarlo.jl:133 for the reference's chain
    assert L.mcio_mcmc_burnin(62500, 1, 3, 3, 1, 0.0) == 0
    assert L.mcio_mcmc_burnin(4000, 8, 3, 3, 1, 0.1) == 400            # floor 64*3 + 16*2*3 = 288 < 400
    assert L.mcio_mcmc_burnin(1000, 8, 12, 5, 1, 0.1) == 928           # the floor is not capped: burn-in steps are extra


def test_resampling_of_carried_mcmc_chains(oracle):
    """mcio_resample_chains (mirror of k_resample_chains): the stored chains of a block, a sample of the finished iteration's target, are
    resampled with probability ~ reweight_new[idx] / reweight_old[idx] (doReweight! has moved the factors, main.jl:322-346) into the
    start population of the next launch -- systematic resampling along the chain order, offset (sqrt 5 - 1) / 2."""
    rng = np.random.default_rng(5)
    # nothing moved, same count: every chain continues itself
    curr = rng.integers(0, 4, size=1000)
    rw = np.array([0.1, 0.2, 0.3, 0.4])
    assert np.array_equal(oracle.resample_chains(curr, rw, rw, 1000), np.arange(1000))
    # nothing moved, twice the chains: every stored chain is continued twice; half the chains: every second one
    assert np.array_equal(np.bincount(oracle.resample_chains(curr, rw, rw, 2000), minlength=1000), np.full(1000, 2))
    half = oracle.resample_chains(curr, rw, rw, 500)
    assert np.all(np.diff(half) == 2)
    # moved factors: the picks are monotone in the chain order, every stored chain is continued floor or ceil of its expected number of
    # times n_new w[curr] / sum(w) (systematic resampling), and a ratio of zero leaves nobody on that integrand
    new = np.array([0.05, 0.4, 0.3, 0.25])
    w = new / rw
    for n_new in (1000, 137, 4096):
        src = oracle.resample_chains(curr, new, rw, n_new)
        assert np.all(np.diff(src) >= 0) and src.min() >= 0 and src.max() < 1000
        expect = n_new * w[curr] / w[curr].sum()
        copies = np.bincount(src, minlength=1000)
        assert np.all(copies >= np.floor(expect - 1e-9)) and np.all(copies <= np.ceil(expect + 1e-9)), n_new
        assert abs(np.bincount(curr[src], minlength=4)[1] / n_new - (np.bincount(curr, minlength=4) * w)[1] / w[curr].sum()) < 0.01 + 0.5 / np.sqrt(n_new)
    src = oracle.resample_chains(curr, np.array([0.0, 0.5, 0.3, 0.2]), rw, 777)
    assert not np.any(curr[src] == 0)
    # all stored chains on one integrand and a chain count that made offset 1/2 an exact tie (16 -> 5: target 2.5 * 16 / 5 = 8): the pick
    # does not depend on the last bit of the ratio
    one = np.zeros(16, dtype=np.int64)
    picks = {tuple(oracle.resample_chains(one, np.array([r, 1.0 - r]), np.array([0.5, 0.5]), 5)) for r in (0.3, 0.3 * (1 + 2e-16), 0.3 * (1 - 2e-16), 0.7)}
    assert len(picks) == 1


def test_weighted_resampling_of_carried_vegasmc_chains(oracle):
    """mcio_resample_weighted (mirror of k_resample_chains' per-chain-weight path): the stored :vegasmc chains of a block are a sample of the
    OLD target density (it contains the map and the reweight factors); the next launch continues stored chain j with probability
    ~ w_j = new target / old target -- systematic resampling along the chain order with one fixed offset, the running sums formed in ONE
    fixed association (256 stretches: sum of the stretches before + running sum along the own stretch)."""
    n = 1000
    assert np.array_equal(oracle.resample_weighted(np.ones(n), n), np.arange(n))                       # nothing moved: every chain goes on
    assert np.array_equal(np.bincount(oracle.resample_weighted(np.ones(n), 3 * n), minlength=n), np.full(n, 3))
    w = np.zeros(n)
    w[[17, 400]] = [1.0, 3.0]
    src = oracle.resample_weighted(w, 400)
    assert set(src) == {17, 400} and np.all(np.diff(src) >= 0) and np.sum(src == 17) == 100           # exactly in proportion
    rng = np.random.default_rng(3)
    w = rng.exponential(size=5000) * (rng.random(5000) < 0.3)                                          # many chains the new target excludes
    for n_new in (50, 5000, 20000):
        src = oracle.resample_weighted(w, n_new)
        assert np.all(np.diff(src) >= 0) and np.all(w[src] > 0.0)                                      # chain order kept, nothing with zero weight continued
        counts = np.bincount(src, minlength=len(w))
        expect = w / w.sum() * n_new
        assert np.all(np.abs(counts - expect) < 1.0 + 1e-9)                                            # systematic: within one of the expected count
    # the association of the running sums, restated: stretches of ceil(n / 256), sums of whole stretches added in order, then the running sum
    per = (len(w) + 255) // 256
    W = np.empty(len(w))
    below = 0.0
    for t in range(256):
        run = 0.0
        for j in range(t * per, min((t + 1) * per, len(w))):
            run += w[j]
            W[j] = below + run
        part = 0.0
        for j in range(t * per, min((t + 1) * per, len(w))):
            part += w[j]
        below += part
    step = W[-1] / 777
    ref = np.array([np.searchsorted(W, (c + 0.6180339887498949) * step, side="right") for c in range(777)])
    assert np.array_equal(oracle.resample_weighted(w, 777), np.minimum(ref, len(w) - 1))


def test_vegasmc_chains_are_carried_while_the_map_stays_and_not_onto_a_first_refinement(oracle):
    """The oracle's mirror of the rule of mci_iteration_run: the chains of a :vegasmc launch on a map train! has never refined go on in the
    next iteration as long as the map stays as it is (adapt = false), and start afresh when train! has refined it in between (their end
    configurations are no sample of the old target yet: profiles/r05_bias.txt A4); out of a launch on a refined map they are carried."""
    def run(carry, train):
        cfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[2], [3]])
        cfg.set_chain_carry(carry)
        out = []
        for it in range(3):
            out.append(cfg.iteration(oracle.VEGASMC, "sphere2", [2.0, 3.0], 4000, 0, 2, it, 11, nchain=8))
            if train:
                cfg.train()
        return out

    keep, fresh = run("auto", False), run("off", False)
    np.testing.assert_array_equal(keep[0], fresh[0])                 # first launches: both start afresh
    assert not np.array_equal(keep[1], fresh[1])                     # the map has not moved: chains go on
    assert not np.array_equal(keep[2], fresh[2])
    keep, fresh = run("auto", True), run("off", True)
    np.testing.assert_array_equal(keep[0], fresh[0])
    np.testing.assert_array_equal(keep[1], fresh[1])                 # refined for the first time in between: afresh
    assert not np.array_equal(keep[2], fresh[2])                     # out of a launch on a refined map: carried (resampled to the moved target)
