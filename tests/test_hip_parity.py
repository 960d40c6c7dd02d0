"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical Philox
streams.  Tolerances (fp64 everywhere; only summation order and libm ulps differ):
    map draw x            : bit-exact
    jac, integrand weight : rel 1e-13
    block sums / packed   : rel 1e-11   (reassociation of up to 1e5-term sums)
    histograms            : rel 1e-9 per bin
    trained grids         : abs 1e-12 * range
    per-iteration mean/std: rel 1e-6 after six train steps (< 1e-3 sigma)
"""
import math

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from catalog_params import bubble_exact, bubble_userdata, genz_exact, genz_userdata

pytestmark = pytest.mark.gpu
PI = math.pi
SEED = 20240229


def ocont(pool=0, lo=0.0, hi=1.0, **kw):
    return dict(kind=0, pool=pool, lower=lo, upper=hi, **kw)


def odisc(pool, lo, hi, **kw):
    return dict(kind=1, pool=pool, lower=lo, upper=hi, **kw)


def hist_split(packed, nobs, ni):
    n = 2 * nobs + 2 + ni + 1
    return packed[:n], packed[n:]


# (name, product Configuration factory, HIP integrand, oracle leaves, oracle dof, oracle builtin, userdata, obs kwargs)
def cases():
    L = math.sqrt(50.0)
    bp = bubble_userdata()
    out = {
        "c1_log_over_sqrt": dict(var=lambda: mci.Continuous(0.0, 1.0), dof=[[1]], f=mci.catalog.log_over_sqrt(),
                                 oleaves=[ocont()], oname="log_over_sqrt", ud=None),
        "c2_gauss16_shared_pool": dict(var=lambda: mci.Continuous(-L, L), dof=[[16]], f=mci.catalog.gaussian(16),
                                       oleaves=[ocont(0, -L, L)], oname="gaussian", ud=[16.0]),
        "c2_gauss4_composite": dict(var=lambda: mci.Continuous([(-L, L)] * 4), dof=[[1]], f=mci.catalog.gaussian(4),
                                    oleaves=[ocont(0, -L, L) for _ in range(4)], oname="gaussian", ud=[4.0]),
        "sphere2_padding": dict(var=lambda: mci.Continuous(0.0, 1.0), dof=[[2], [3]], f=mci.catalog.sphere2(),
                                oleaves=[ocont()], oname="sphere2", ud=None),
        "hypersphere": dict(var=lambda: mci.Continuous(-1.0, 1.0), dof=[[2], [3], [4]], f=mci.catalog.hypersphere(3),
                            oleaves=[ocont(0, -1.0, 1.0)], oname="hypersphere", ud=[3.0]),
        # BASELINE configs[4] (C5): 4 integrals sharing a 12-D Continuous pool, nested dof -> padding_probability (variable.jl:628-641)
        "c5_nested_gauss": dict(var=lambda: mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], f=mci.catalog.nested_gauss(),
                                oleaves=[ocont()], oname="nested_gauss", ud=[4.0, 3.0, 6.0, 9.0, 12.0]),
        "discrete": dict(var=lambda: mci.Discrete(1, 3), dof=[[1]], f=mci.catalog.discrete_id(),
                         oleaves=[odisc(0, 1, 3)], oname="discrete_id", ud=None),
        "discrete2_composite": dict(var=lambda: mci.Discrete([(1, 3), (1, 4)]), dof=[[1]], f=mci.catalog.one(),
                                    oleaves=[odisc(0, 1, 3), odisc(0, 1, 4)], oname="one", ud=None),
        "singular2_composite": dict(var=lambda: mci.Continuous([(0.0, PI)] * 3), dof=[[1]], f=mci.catalog.singular2(),
                                    oleaves=[ocont(0, 0.0, PI) for _ in range(3)], oname="singular2", ud=None),
        "bubble": dict(var=lambda: (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0),
                                    mci.Continuous(0.0, 2 * PI, alpha=3.0), mci.Continuous(0.0, bp[1], alpha=3.0),
                                    mci.Discrete(1, 4, adapt=False)),
                       dof=[[1, 1, 1, 1, 1]], f=mci.catalog.bubble(), obs=[np.zeros(4)], measure=mci.bin_by(4),
                       oleaves=[ocont(0, 0, 1, alpha=3.0), ocont(1, 0, PI, alpha=3.0), ocont(2, 0, 2 * PI, alpha=3.0),
                                ocont(3, 0, bp[1], alpha=3.0), odisc(4, 1, 4, adapt=False)],
                       oname="bubble", ud=bp, obs_nbin=[4], obs_bin_draw=[4]),
    }
    return out


CASES = cases()


def make(name, oracle):
    c = CASES[name]
    cfg = mci.Configuration(var=c["var"](), dof=c["dof"], obs=c.get("obs"), seed=SEED)
    eng = mci.Engine(cfg, c["f"], measure=c.get("measure"))
    ocfg = oracle.Config(c["oleaves"], c["dof"], obs_nbin=c.get("obs_nbin"), obs_bin_draw=c.get("obs_bin_draw"))
    return c, cfg, eng, ocfg


@pytest.mark.parametrize("name", list(CASES))
def test_map_draw_and_integrand_match_oracle(oracle, name):
    """rows a2/a3/a4: inverse-CDF draw (sampler.jl:293-305, :13-22, :410-418) -- x bit-exact."""
    c, cfg, eng, ocfg = make(name, oracle)
    n, npb = 4096, 5000
    x, jac, w = eng.sample_dump(n, nevalperblock=npb, block_index=3, iteration=2, seed=SEED)
    oc = ocfg.c
    fn = oracle.builtin(c["oname"])
    import ctypes as C
    ud = np.ascontiguousarray(c["ud"] if c["ud"] is not None else [0.0], dtype=np.float64)
    wo = np.zeros(oc.Ni)
    for s in range(0, n, 97):
        gs = 3 * npb + s
        k = 0
        jaco = 1.0
        xo = np.zeros(oc.ndraw)
        for vi in range(oc.npool):
            nl = oc.pool_nleaf[vi]
            for idx in range(1, oc.maxdof[vi] + 1):
                us = [oracle.uniform(SEED, 2 * 8 + 0, gs, k + l) for l in range(nl)]
                ocfg.pool_create(vi, idx, us)
                jaco /= np.ctypeslib.as_array(oc.pool_prob[vi], shape=(idx + 1,))[idx]
                for l in range(nl):
                    xo[k + l] = ocfg.pool_data(oc.pool_leaf0[vi] + l)[idx - 1]
                k += nl
        assert np.array_equal(x[s], xo), (name, s, x[s], xo)
        assert jac[s] == pytest.approx(jaco, rel=1e-13)
        C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)(fn)(xo.ctypes.data, wo.ctypes.data, ud.ctypes.data)
        np.testing.assert_allclose(w[s], wo, rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("name", list(CASES))
def test_vegas_iteration_packed_matches_oracle(oracle, name):
    """rows a6/a7/a8/a11/a12: one iteration of blocks -> [obsSum|obsSqSum|normalization|neval|visited|hist]."""
    c, cfg, eng, ocfg = make(name, oracle)
    block, npb = 8, 4000
    got = eng.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)
    ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
    gs, gh = hist_split(got, eng.nobs, cfg.N)
    rs, rh = hist_split(ref, eng.nobs, cfg.N)
    np.testing.assert_allclose(gs, rs, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(gh, rh, rtol=1e-9)
    assert got[2 * eng.nobs + 1] == block * npb  # neval


@pytest.mark.parametrize("name", ["c2_gauss16_shared_pool", "c5_nested_gauss", "bubble", "singular2_composite"])
def test_vegas_32_bit_stream_matches_oracle(oracle, name):
    """the opt-in cheaper stream of :vegas (mci_set_rng_bits(32): one Philox word per draw, four draws per block): draws bit-exact,
    iteration sums at the usual tolerances, and a whole run on the oracle's trajectory"""
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_rng_bits(32)
    ocfg.set_rng_bits(32)
    x, jac, w = eng.sample_dump(512, nevalperblock=3000, block_index=2, iteration=1, seed=SEED)
    oc = ocfg.c
    for s in range(0, 512, 37):
        gs = 2 * 3000 + s
        k = 0
        xo = np.zeros(oc.ndraw)
        for vi in range(oc.npool):
            nl = oc.pool_nleaf[vi]
            for idx in range(1, oc.maxdof[vi] + 1):
                us = [oracle.uniform(SEED, 1 * 8 + 0, gs, k + l, bits=32) for l in range(nl)]
                ocfg.pool_create(vi, idx, us)
                for l in range(nl):
                    xo[k + l] = ocfg.pool_data(oc.pool_leaf0[vi] + l)[idx - 1]
                k += nl
        assert np.array_equal(x[s], xo), (name, s)
    block, npb = 8, 4000
    got = eng.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)
    ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
    gs_, gh = hist_split(got, eng.nobs, cfg.N)
    rs, rh = hist_split(ref, eng.nobs, cfg.N)
    np.testing.assert_allclose(gs_, rs, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(gh, rh, rtol=1e-9)
    plain = mci.Engine(mci.Configuration(var=c["var"](), dof=c["dof"], obs=c.get("obs"), seed=SEED), c["f"], measure=c.get("measure"))
    assert not np.allclose(plain.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)[:eng.nobs], got[:eng.nobs], rtol=1e-9)   # another stream
    eng.set_train_walk("serial")
    r = eng.integrate("vegas", neval=40000, niter=5, block=16, seed=SEED)
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=40000, niter=5, block=16, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-6, atol=1e-300)


@pytest.mark.parametrize("name", ["c2_gauss16_shared_pool", "c5_nested_gauss", "bubble"])
def test_seven_round_philox_stream_matches_oracle(oracle, name):
    """the opt-in cheaper generator (mci_set_rng_rounds(7): Philox4x32-7 for every stream, pinned on the Random123 vectors in
    tests/golden): draws bit-exact, one iteration of each solver at the usual tolerances, a whole :vegas run on the oracle's
    trajectory, and a stream that differs from the ten-round one"""
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_rng_rounds(7)
    oracle.set_rng_rounds(7)
    try:
        x, jac, w = eng.sample_dump(512, nevalperblock=3000, block_index=2, iteration=1, seed=SEED)
        oc = ocfg.c
        for s in range(0, 512, 37):
            gs = 2 * 3000 + s
            k = 0
            xo = np.zeros(oc.ndraw)
            for vi in range(oc.npool):
                nl = oc.pool_nleaf[vi]
                for idx in range(1, oc.maxdof[vi] + 1):
                    us = [oracle.uniform(SEED, 1 * 8 + 0, gs, k + l) for l in range(nl)]
                    ocfg.pool_create(vi, idx, us)
                    for l in range(nl):
                        xo[k + l] = ocfg.pool_data(oc.pool_leaf0[vi] + l)[idx - 1]
                    k += nl
            assert np.array_equal(x[s], xo), (name, s)
        block, npb = 8, 4000
        got = eng.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)
        ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
        gs_, gh = hist_split(got, eng.nobs, cfg.N)
        rs, rh = hist_split(ref, eng.nobs, cfg.N)
        np.testing.assert_allclose(gs_, rs, rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(gh, rh, rtol=1e-9)
        plain = mci.Engine(mci.Configuration(var=c["var"](), dof=c["dof"], obs=c.get("obs"), seed=SEED), c["f"], measure=c.get("measure"))
        assert not np.allclose(plain.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)[:eng.nobs], got[:eng.nobs], rtol=1e-9)   # another stream
        for solver, osolver, kw in (("vegasmc", oracle.VEGASMC, dict(nchain=64)), ("mcmc", oracle.MCMC, dict(nchain=16))):
            c2, cfg2, eng2, ocfg2 = make(name, oracle)
            eng2.set_rng_rounds(7)
            if solver == "mcmc":
                ocfg2.set_thermal_ratio(0.1)
                kw2 = dict(thermal_ratio=0.1, **kw)
            else:
                kw2 = kw
            got = eng2.iteration(solver, 3200, 0, 4, iteration=1, seed=SEED, **kw2)
            ref = ocfg2.iteration(osolver, c["oname"], c["ud"], 3200, 0, 4, 1, SEED, **kw)
            gs_, gh = hist_split(got, eng2.nobs, cfg2.N)
            rs, rh = hist_split(ref, eng2.nobs, cfg2.N)
            np.testing.assert_allclose(gs_, rs, rtol=1e-9, atol=1e-300, err_msg=solver)
            np.testing.assert_allclose(gh, rh, rtol=1e-8, err_msg=solver)
        eng.set_train_walk("serial")
        r = eng.integrate("vegas", neval=40000, niter=5, block=16, seed=SEED)
        o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=40000, niter=5, block=16, seed=SEED)
        np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-6, atol=1e-300)
        if name == "c2_gauss16_shared_pool":
            # both opt-in streams: the histogram-copy rule switches to sixteen copies in one 1024-thread workgroup per CU
            c3, cfg3, eng3, ocfg3 = make(name, oracle)
            eng3.set_rng_rounds(7)
            eng3.set_rng_bits(32)
            ocfg3.set_rng_bits(32)
            got = eng3.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)
            assert eng3.histogram_copies() == 16 and eng3.kernel_times_ms(1)[2] == 1024
            ref = ocfg3.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
            gs_, gh = hist_split(got, eng3.nobs, cfg3.N)
            rs, rh = hist_split(ref, eng3.nobs, cfg3.N)
            np.testing.assert_allclose(gs_, rs, rtol=1e-11, atol=1e-300)
            np.testing.assert_allclose(gh, rh, rtol=1e-9)
    finally:
        oracle.set_rng_rounds(10)


@pytest.mark.parametrize("threads", [None, 512], ids=["plan_a", "plan_b_512"])
def test_c4_32_bit_stream_with_gather_phase_matches_oracle(oracle, threads):
    """32 grids (split-all pass with the dimension-major gather phase, 768 or 512 threads) on the 32-bit stream: 8 Philox blocks per
    sample instead of 16"""
    ud = genz_userdata(32)
    cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.genz_product_peak(32), rng_bits=32, **(dict(threads=threads) if threads else {}))
    ocfg = oracle.Config([ocont(0) for _ in range(32)], [[1]])
    ocfg.set_rng_bits(32)
    got = eng.iteration("vegas", 2000, 0, 4, iteration=0, seed=SEED)
    ref = ocfg.iteration(oracle.VEGAS, "genz_product_peak", ud, 2000, 0, 4, 0, SEED)
    np.testing.assert_allclose(got[:4], ref[:4], rtol=1e-11)
    np.testing.assert_allclose(got[4:], ref[4:], rtol=1e-9)


@pytest.mark.parametrize("chunk,layout,measurefreq,keep_tile0", [(1000, "c4", 1, 0), (37, "c4", 1, 0), (4000, "c4", 1, 1), (10**9, "c4", 1, 0), (500, "three_grids", 3, 0)],
                         ids=["1000", "tiny_ragged", "tile0_in_pass", "one_chunk", "three_grids_mf3"])
def test_many_grid_launch_in_chunks_matches_oracle(oracle, overrides, chunk, layout, measurefreq, keep_tile0):
    """The reference's loop is constant memory in neval (vegas/montecarlo.jl:117-187).  A many-grid launch (BASELINE configs[3]: 32 grids,
    the histograms in two LDS tiles) parks (weights, bins) per sample for the replay; it runs in chunks of the blocks' samples -- sample
    pass -> replay per chunk, same Philox indices, partial rows accumulated -- so that the parked stream is bounded.  With the chunk forced
    to a few samples the launch equals the oracle at the tolerances of the one-chunk launch: a block range, a ragged last chunk, tile 0
    kept in the sample pass, and (on a small three-tile layout whose any-cadence kernel compiles in a second) measurefreq = 3."""
    overrides.set("split_chunk", chunk)
    if keep_tile0:
        overrides.set("no_split_all", 1)
    if layout == "c4":
        ud, ndraw = genz_userdata(32), 32
        cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=SEED)
        eng = mci.Engine(cfg, mci.catalog.genz_product_peak(32))
        ocfg, oname = oracle.Config([ocont(0) for _ in range(32)], [[1]]), "genz_product_peak"
    else:
        overrides.set("table_mode", 3)
        overrides.set("hist_tile_bins", 1000)          # one 999-bin leaf per tile -> 3 tiles, every one replayed
        ud, ndraw = None, 3
        cfg = mci.Configuration(var=mci.Continuous([(0.0, PI)] * 3), dof=[[1]], seed=SEED)
        eng = mci.Engine(cfg, mci.catalog.singular2())
        ocfg, oname = oracle.Config([ocont(0, 0.0, PI) for _ in range(3)], [[1]]), "singular2"
    npb, lo, hi = 2003, 1, 5                           # (an odd block length: the last chunk is ragged)
    got = eng.iteration("vegas", npb, lo, hi, iteration=2, seed=SEED, measurefreq=measurefreq)
    ref = ocfg.iteration(oracle.VEGAS, oname, ud, npb, lo, hi, 2, SEED, measurefreq=measurefreq)
    nchunks, held = eng.split_chunks()
    per = max(4, (chunk // (hi - lo)) & ~3)
    assert nchunks == (1 if per >= npb else -(-npb // per)), (nchunks, per)
    tdraws = ndraw // 2 if keep_tile0 else ndraw       # the replayed draws' bins, 10 bits each (999-bin grids), next to 8 B of weight per parked sample
    assert held == (hi - lo) * min(per, npb) * (8 + 4 * ((tdraws * 10 + 31) // 32))      # C4: 48 B per sample
    np.testing.assert_allclose(got[:4], ref[:4], rtol=1e-11)
    np.testing.assert_allclose(got[4:], ref[4:], rtol=1e-9)
    # a second launch on the same problem with another chunking: the rows of the first are not carried into it
    overrides.set("split_chunk", 3 * chunk if chunk < 10**9 else 500)
    got2 = eng.iteration("vegas", npb, lo, hi, iteration=2, seed=SEED, measurefreq=measurefreq)
    np.testing.assert_allclose(got2, got, rtol=1e-12)


def test_vegas_iteration_block_range_and_measurefreq(oracle):
    """blocks [lo,hi) are the MPI-rank partition (main.jl:152-166); measurefreq (vegas/montecarlo.jl:148)."""
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    got = eng.iteration("vegas", 3000, 4, 8, iteration=1, seed=SEED, measurefreq=3)
    ref = ocfg.iteration(oracle.VEGAS, "sphere2", None, 3000, 4, 8, 1, SEED, measurefreq=3)
    np.testing.assert_allclose(got, ref, rtol=1e-9)
    assert got[2 * eng.nobs] == pytest.approx(4 * 1000 + 5e-10, rel=1e-13)  # normalization = measured samples + 1e-10 per block + 1e-10


@pytest.mark.parametrize("walk", ["serial", "serial_general", "prefix"])
@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "c2_gauss16_shared_pool", "discrete", "bubble", "singular2_composite", "c5_nested_gauss"])
def test_train_matches_oracle(oracle, name, walk, overrides):
    """rows a9/a10: smooth -> rescale -> refine (variable.jl:206-239), Discrete (:369-382).  `serial` walks the reference's recurrence
    with the decisions of the prefix-scan form given and checked (falling back to `serial_general`, the recurrence with its compares
    and branches, where one does not hold or a bin yields several points); `prefix` is the scan + bisection form."""
    overrides.set("train_walk", {"serial": 1, "serial_general": 2, "prefix": 0}[walk])
    c, cfg, eng, ocfg = make(name, oracle)
    block, npb = 8, 4000
    eng.run("vegas", npb, 0, block, 0, SEED)
    m, e = eng.finish("vegas", block, adapt=True)
    packed = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
    ocfg.train()
    om, oe = oracle.mean_std(packed[:eng.nobs], packed[eng.nobs:2 * eng.nobs], block)
    np.testing.assert_allclose(m, om, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(e, oe, rtol=1e-8, atol=1e-300)
    for i, lf in enumerate(c["oleaves"]):
        if lf["kind"] == 0:
            g, og = eng.grid(i), ocfg.grid(i)
            assert g[0] == og[0] and g[-1] == og[-1]
            np.testing.assert_allclose(g, og, rtol=0, atol=1e-12 * (lf["upper"] - lf["lower"]))
            assert np.all(np.diff(g) > 0)
        else:
            d, a = eng.distribution(i)
            np.testing.assert_allclose(d, ocfg.distribution(i), rtol=1e-11)
            np.testing.assert_allclose(a, ocfg.accumulation(i), rtol=1e-11)


@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "sphere2_padding", "c2_gauss16_shared_pool"])
def test_mid_size_launch_spreads_its_atomic_flush_over_three_buffers(oracle, name):
    """launches of 65..256 partial rows (a few 1e5 samples) add their LDS histograms to THREE merged-histogram buffers, row r to buffer
    r % 3 (256 rows on one buffer are 256 serialized atomics per bin), and the merge sums and clears all three: one iteration against the
    oracle at the usual tolerances, and the same iteration again -- the buffers were left clean"""
    c, cfg, eng, ocfg = make(name, oracle)
    block, npb = 16, 40000
    got = eng.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)
    ms, wg, th = eng.kernel_times_ms(1)
    assert 64 < wg <= 256, wg
    ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
    gs, gh = hist_split(got, eng.nobs, cfg.N)
    rs, rh = hist_split(ref, eng.nobs, cfg.N)
    np.testing.assert_allclose(gs, rs, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(gh, rh, rtol=1e-9)
    again = eng.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)
    np.testing.assert_allclose(again, got, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("walk", ["serial", "serial_general", "prefix"])
@pytest.mark.parametrize("ninc", [1025, 1026, 2500])
def test_train_of_a_grid_longer_than_julias_simd_block_matches_oracle(oracle, ninc, walk, overrides):
    """train! on grids of 1024 / 1025 / 2499 increments: Julia's sum() (common.jl:72, variable.jl:226) runs its @simd block up to 1024
    elements and splits longer vectors pairwise at the midpoint first (base/reduce.jl mapreduce_impl) -- mcio_sum_julia in the oracle,
    sum_julia on the device; same tolerances as the default 999 increments"""
    overrides.set("train_walk", {"serial": 1, "serial_general": 2, "prefix": 0}[walk])
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0, ninc=ninc), dof=[[2]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.x2y2())
    ocfg = oracle.Config([ocont(npts=ninc)], [[2]])
    block, npb = 8, 8000
    eng.run("vegas", npb, 0, block, 0, SEED)
    eng.finish("vegas", block, adapt=True)
    ocfg.iteration(oracle.VEGAS, "x2y2", None, npb, 0, block, 0, SEED)
    ocfg.train()
    g, og = eng.grid(0), ocfg.grid(0)
    assert len(g) == ninc and g[0] == og[0] and g[-1] == og[-1] and np.all(np.diff(g) > 0)
    np.testing.assert_allclose(g, og, rtol=0, atol=1e-12)


@pytest.mark.parametrize("walk", ["serial", "serial_general", "prefix"])
@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "sphere2_padding", "bubble", "c2_gauss4_composite"])
def test_full_integrate_matches_oracle(oracle, name, walk, overrides):
    """rows a13-a15: the whole loop inside the library (mci_integrate) vs the oracle's loop, same seed.

    The adaptation loop grid -> histogram -> grid amplifies rounding-level differences by ~1-2 orders of
    magnitude per train! step (measured: 1 ulp of device pow/log becomes 1e-6 after six steps), so the run-level
    tolerance depends on how the refinement walk rounds: `serial` = the reference's recurrence order
    (variable.jl:227-234; override train_walk = 1), `prefix` = the default scan + bisection form (one train! step of
    either agrees with the oracle to 1e-12 of the range, test_train_matches_oracle)."""
    overrides.set("train_walk", {"serial": 1, "serial_general": 2, "prefix": 0}[walk])
    rtol, sig = (1e-6, 1e-3) if walk != "prefix" else (1e-4, 5e-2)
    c, cfg, eng, ocfg = make(name, oracle)
    r = eng.integrate("vegas", neval=40000, niter=6, block=16, seed=SEED)
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=40000, niter=6, block=16, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=rtol, atol=1e-300)
    np.testing.assert_allclose(r["iter_std"], o["iter_std"], rtol=100 * rtol, atol=1e-300)
    np.testing.assert_allclose(r["mean"], o["mean"], rtol=rtol)
    np.testing.assert_allclose(r["stdev"], o["stdev"], rtol=100 * rtol)
    np.testing.assert_allclose(r["chi2"], o["chi2"], rtol=1000 * rtol, atol=1e-9)
    assert np.all(np.abs(r["mean"] - o["mean"]) < sig * o["stdev"])


@pytest.mark.parametrize("case", ["adapting_peak", "flat", "gauss16"])
def test_serial_walk_with_given_decisions_is_the_recurrence_bit_for_bit(case):
    """The serial walk of train! (variable.jl:227-234) takes the recurrence's decisions -- does this bin yield a new grid point -- from
    the prefix-scan form, walks the additions and subtractions alone and checks every decision against the exact record; a wrong one
    (acc_f within rounding of f_ninc) or a bin with several points sends it through the general form.  Either way the grid is the
    recurrence's: deterministic runs (bit-reproducible histograms) under "serial" and "serial_general" give IDENTICAL grids and
    iterations -- a narrow peak met by a uniform grid (many points per bin at first), a flat integrand (every bin is a tie: acc_f =
    f_ninc up to rounding) and the headline layout; and with one decision deliberately wrong (a test hook) the check catches it and
    the general form's result comes out."""
    import math
    out = []
    for walk in ("serial", "serial_general", "serial_wrong_decision"):
        if case == "gauss16":
            L = math.sqrt(50.0)
            cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=SEED)
            eng = mci.Engine(cfg, mci.catalog.gaussian(16), deterministic=True)
        else:
            cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]], seed=SEED)
            body = "w[0] = 1.0;" if case == "flat" else "const double t = (x[0] - 0.3) * 400.0; w[0] = exp(-t * t);"
            eng = mci.Engine(cfg, mci.Integrand(body), deterministic=True)
        eng.set_train_walk(walk)
        r = eng.integrate("vegas", neval=200000, niter=8, block=16, seed=SEED)
        out.append((r["iter_mean"].copy(), r["iter_std"].copy(), eng.grid(0), eng.walk_counts()))
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1])
        assert np.array_equal(out[0][2], other[2])
    assert np.all(np.diff(out[0][2]) > 0)
    # (slots, general form): one walk per iteration; the hook's wrong decision is caught every time and the walk redone in the general form
    assert sum(out[0][3]) == 8 and out[1][3] == (0, 8) and out[2][3] == (0, 8)
    if case != "flat":
        assert out[0][3] == (8, 0)


@pytest.mark.parametrize("case_id", [0, 1, 3, 6, 9, 14])
def test_serial_walk_slots_against_the_general_form_on_random_layouts(case_id):
    """a few cases of `tools/fuzz_layouts.py --walk` (tests/layout_cases.py check_walks_agree; the campaign's record is in profiles/)"""
    from layout_cases import check_walks_agree
    check_walks_agree(case_id, SEED)


def test_full_integrate_default_walk_at_large_launches_is_the_reference_recurrence(oracle):
    """The DEFAULT refinement walk: once an iteration's sample launch is long enough to hide its ~14 us (>= 2^26 samples per
    rank: the headline configuration's 1e8 included) train! runs the reference's serial recurrence (variable.jl:227-234), so whole
    runs agree with the oracle at the 1e-6 level of the serial walk; below that size the prefix-scan form keeps the launch-bound regime at 50 us per iteration
    (1e-4 level, test_full_integrate_matches_oracle[prefix]).  Engine.set_train_walk("serial") asks for the recurrence at any size."""
    neval = (1 << 26) + 16
    c, cfg, eng, ocfg = make("c1_log_over_sqrt", oracle)
    r = eng.integrate("vegas", neval=neval, niter=4, block=16, seed=SEED)
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=neval, niter=4, block=16, seed=SEED, nthreads=16)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-6, atol=1e-300)
    np.testing.assert_allclose(r["iter_std"], o["iter_std"], rtol=1e-4, atol=1e-300)
    np.testing.assert_allclose(eng.grid(0), ocfg.grid(0), rtol=0, atol=1e-8)
    # the explicit knob, at a small size: same 1e-6 level as the override train_walk = 1
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    eng.set_train_walk("serial")
    r = eng.integrate("vegas", neval=40000, niter=6, block=16, seed=SEED)
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=40000, niter=6, block=16, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-6, atol=1e-300)


@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "sphere2_padding", "bubble", "discrete2_composite", "c5_nested_gauss"])
def test_vegasmc_iteration_matches_oracle(oracle, name):
    """row a16: many-chain VegasMC block vs the oracle run with the same chain decomposition."""
    c, cfg, eng, ocfg = make(name, oracle)
    block, npb, nchain = 4, 6400, 64
    got = eng.iteration("vegasmc", npb, 0, block, iteration=0, seed=SEED, nchain=nchain)
    ref = ocfg.iteration(oracle.VEGASMC, c["oname"], c["ud"], npb, 0, block, 0, SEED, nchain=nchain)
    gs, gh = hist_split(got, eng.nobs, cfg.N)
    rs, rh = hist_split(ref, eng.nobs, cfg.N)
    np.testing.assert_allclose(gs, rs, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(gh, rh, rtol=1e-8)
    # propose[2, 1, vi] / accept[2, 1, vi] (vegas_mc/updates.jl:90-92), everything else at its clearStatistics! offset
    pr, ac = eng.acceptance()
    nd, m = cfg.N + 1, max(cfg.N + 1, len(cfg.var))
    npa = 3 * nd * m
    np.testing.assert_allclose(pr.ravel(), ref[-2 * npa:-npa], rtol=1e-12)
    np.testing.assert_allclose(ac.ravel(), ref[-npa:], rtol=1e-12)
    live = pr[1, 0, :len(cfg.var)]
    assert live.sum() > 0.5 * block * npb and abs(pr.sum() - live.sum()) < 1e-4


def test_vegasmc_single_chain_is_the_reference_chain(oracle):
    """nchain = 1: exactly the reference's one-chain-per-block Markov chain (vegas_mc/montecarlo.jl:184)."""
    c, cfg, eng, ocfg = make("c1_log_over_sqrt", oracle)
    got = eng.iteration("vegasmc", 3000, 0, 2, iteration=0, seed=SEED, nchain=1)
    ref = ocfg.iteration(oracle.VEGASMC, "log_over_sqrt", None, 3000, 0, 2, 0, SEED, nchain=1)
    np.testing.assert_allclose(got, ref, rtol=1e-9)


@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "sphere2_padding", "hypersphere", "bubble", "discrete2_composite", "singular2_composite",
                                  "c5_nested_gauss"])
@pytest.mark.parametrize("nchain", [1, 16])
def test_mcmc_iteration_matches_oracle(oracle, name, nchain):
    """row f1: the :mcmc chains (mcmc/montecarlo.jl:72-184, mcmc/updates.jl) against the oracle on the same
    Philox streams; nchain = 1 is exactly the reference's one-chain-per-block walk over (integrand, variables)."""
    c, cfg, eng, ocfg = make(name, oracle)
    block, npb = 4, 3200
    got = eng.iteration("mcmc", npb, 0, block, iteration=2, seed=SEED, nchain=nchain, thermal_ratio=0.1)
    ocfg.set_thermal_ratio(0.1)
    ref = ocfg.iteration(oracle.MCMC, c["oname"], c["ud"], npb, 0, block, 2, SEED, nchain=nchain)
    gs, gh = hist_split(got, eng.nobs, cfg.N)
    rs, rh = hist_split(ref, eng.nobs, cfg.N)
    np.testing.assert_allclose(gs, rs, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(gh, rh, rtol=1e-9)   # unit weights: counts + the 1e-10 clearStatistics offsets
    # config.propose / config.accept [3][Nd][max(Nd, Nv)] (configuration.jl:185-186; mcmc/updates.jl:48,100,138): the tail of the
    # packed buffer, integer counts + clearStatistics offsets -> equal to the oracle's entry by entry
    pr, ac = eng.acceptance()
    nd, m = cfg.N + 1, max(cfg.N + 1, len(cfg.var))
    npa = 3 * nd * m
    np.testing.assert_allclose(pr.ravel(), ref[-2 * npa:-npa], rtol=1e-12)
    np.testing.assert_allclose(ac.ravel(), ref[-npa:], rtol=1e-12)
    assert pr.shape == (3, nd, m) and 0.05 * block * npb < pr.sum() - npa * (block + 1) * 1e-8 and np.all(ac <= pr)
    assert pr[0, nd - 1].sum() > 1.0 and pr[1, nd - 1].sum() < 1e-6   # the normalisation integrand jumps, but has no variable to change
    # the holding-time diagnostic behind the automatic chain length: integer bookkeeping on the same accept decisions
    hh = eng.hold_histogram()
    np.testing.assert_array_equal(hh, ocfg.hold_hist)
    assert hh.sum() == block * nchain


def test_mcmc_custom_neighbor_graph_and_measurefreq(oracle):
    """`neighbor` kwarg (configuration.jl:211-221: undirected 1-based edge list) and measurefreq (mcmc/montecarlo.jl:144)."""
    cfg = mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[2], [3], [4]], neighbor=[(1, 4), (1, 2), (1, 3), (2, 3)], seed=SEED)
    assert cfg.neighbor_lists() == [[1, 2, 3], [0, 2], [0, 1], [0]]
    eng = mci.Engine(cfg, mci.catalog.hypersphere(3))
    ocfg = oracle.Config([ocont(0, -1.0, 1.0)], [[2], [3], [4]])
    ocfg.set_neighbor(cfg.neighbor_lists())
    ocfg.set_thermal_ratio(0.25)
    got = eng.iteration("mcmc", 4000, 0, 2, iteration=0, seed=SEED, nchain=4, measurefreq=3, thermal_ratio=0.25)
    ref = ocfg.iteration(oracle.MCMC, "hypersphere", [3.0], 4000, 0, 2, 0, SEED, nchain=4, measurefreq=3)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)


def test_mcmc_full_integrate_matches_oracle(oracle):
    """the whole :mcmc loop (doReweight! main.jl:183, train!, Result) inside the library vs the oracle's loop."""
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    goal = [1.0, 2.0, 1.0]
    r = eng.integrate("mcmc", neval=32000, niter=5, block=8, seed=SEED, nchain=2, reweight_goal=goal)
    ocfg.set_reweight_goal(goal)
    o = ocfg.integrate(oracle.MCMC, "sphere2", None, neval=32000, niter=5, block=8, seed=SEED, nchain=2)
    # accept/reject decisions are discrete: rounding-level grid differences either leave an iteration
    # bit-compatible or change a decision; five iterations at this size stay on the same trajectory
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-6)
    np.testing.assert_allclose(r["iter_std"], o["iter_std"], rtol=1e-4)
    np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=1e-9)


@pytest.mark.parametrize("name", ["sphere2_padding", "c5_nested_gauss", "bubble"])
def test_vegasmc_full_integrate_matches_oracle(oracle, name):
    """the whole :vegasmc loop inside the library -- chains, block merge, DEVICE doReweight! (main.jl:183, :322-346) with a
    reweight_goal, train!, Result -- against the oracle's loop over several iterations: the reweight vector the next
    iteration's chains see comes from the device-side twin of doReweight!, so its parity is what keeps iterations 2.. on the
    oracle's trajectory."""
    c, cfg, eng, ocfg = make(name, oracle)
    ni = cfg.N
    goal = [1.0 + 0.5 * i for i in range(ni + 1)]
    r = eng.integrate("vegasmc", neval=48000, niter=5, block=8, seed=SEED, nchain=4, reweight_goal=goal)
    ocfg.set_reweight_goal(goal)
    o = ocfg.integrate(oracle.VEGASMC, c["oname"], c["ud"], neval=48000, niter=5, block=8, seed=SEED, nchain=4)
    assert np.all(np.isfinite(r["iter_mean"]))
    # accept/reject decisions are discrete: a rounding-level difference of a trained grid (the histogram's sums are added in another
    # order than the oracle's sequential chain adds them) can flip one and move a chain; the first iterations stay on the oracle's
    # trajectory to rounding -- which is what pins the device-side doReweight! between them --, the later ones and the factors at
    # the end must at least agree statistically
    np.testing.assert_allclose(r["iter_mean"][:2], o["iter_mean"][:2], rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose(r["iter_std"][:2], o["iter_std"][:2], rtol=1e-5, atol=1e-300)
    assert np.all(np.abs(r["iter_mean"] - o["iter_mean"]) <= 6 * np.hypot(r["iter_std"], o["iter_std"]) + 1e-300)
    np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=5e-3)
    # ... and with one lane per chain and the reference's own chain count the factors themselves stay on it for two whole iterations
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_chain_speculation(1)
    r = eng.integrate("vegasmc", neval=48000, niter=2, block=8, seed=SEED, nchain=4, reweight_goal=goal)
    ocfg.set_reweight_goal(goal)
    o = ocfg.integrate(oracle.VEGASMC, c["oname"], c["ud"], neval=48000, niter=2, block=8, seed=SEED, nchain=4)
    np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=1e-7)


@pytest.mark.parametrize("nchain", [1, 0])
def test_c5_mcmc_full_integrate(oracle, nchain):
    """BASELINE configs[4] through mci_integrate: 4 nested Gaussians on a 12-D pool, solver = :mcmc (mcmc/montecarlo.jl:72-184;
    padding variable.jl:628-641), with the reference's one chain per block (nchain = 1: same-stream parity with the oracle's
    loop) and with the automatic many-chain setting (nchain = 0: no oracle twin of the measured chain length -- the estimate
    must sit on the exact products of erf, main.jl:322-346 reweighting included)."""
    c, cfg, eng, ocfg = make("c5_nested_gauss", oracle)
    exact = np.array([math.erf(5.0) ** d for d in (3, 6, 9, 12)])
    if nchain == 1:
        r = eng.integrate("mcmc", neval=64000, niter=4, block=8, seed=SEED, nchain=1)
        o = ocfg.integrate(oracle.MCMC, "nested_gauss", c["ud"], neval=64000, niter=4, block=8, seed=SEED, nchain=1)
        np.testing.assert_allclose(r["iter_mean"][:2], o["iter_mean"][:2], rtol=1e-7)
        assert np.all(np.abs(r["iter_mean"] - o["iter_mean"]) <= 6 * np.hypot(r["iter_std"], o["iter_std"]) + 1e-300)
        np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=1e-3)
    else:
        eng.integrate("mcmc", neval=4 * 10**6, niter=5, block=16, seed=SEED)
        r = eng.integrate("mcmc", neval=4 * 10**6, niter=10, block=16, seed=SEED, first_iteration=5, ignore=0)
        assert np.all(np.abs(r["mean"] - exact) < 5 * r["stdev"]) and np.all(r["stdev"] < 0.02), (r["mean"], r["stdev"])


COMPLEX_BODY = "w[0] = x[0]; w[1] = 0.0; w[2] = 0.5 * x[0]; w[3] = x[0] * x[0];"   # TestComplex2 (test/montecarlo.jl:172-185) + a mixed one
MEASURE_BODY = "if (idx < 0 || idx == 0) obs_add(0, rw[0]); if (idx < 0 || idx == 1) { obs_add(1, rw[1]); obs_add(2, rw[1] * 2.0); }"   # Sphere3 (:71-84)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_complex_weights_match_oracle(oracle, solver):
    """type=ComplexF64 -- abs() is the modulus, (re, im) are separate statistics columns
    (main.jl:279,284,302-305; statistics.jl:207-214)."""
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1], [1]], type=complex, seed=SEED)
    eng = mci.Engine(cfg, mci.Integrand(COMPLEX_BODY))
    assert eng.nobs == 4
    ocfg = oracle.Config([ocont()], [[1], [1]], obs_nbin=[2, 2])
    ocfg.set_ncomp(2)
    fn = oracle.compile_c_integrand(COMPLEX_BODY)
    osolver = dict(vegas=oracle.VEGAS, vegasmc=oracle.VEGASMC, mcmc=oracle.MCMC)[solver]
    got = eng.iteration(solver, 4000, 0, 4, iteration=0, seed=SEED, nchain=8)
    ref = ocfg.iteration(osolver, fn, None, 4000, 0, 4, 0, SEED, nchain=8)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    r = eng.integrate(solver, neval=40000, niter=4, block=8, seed=SEED, nchain=4)
    o = ocfg.integrate(osolver, fn, None, neval=40000, niter=4, block=8, seed=SEED, nchain=4)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-5, atol=1e-300)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_user_measure_matches_oracle(oracle, solver):
    """a user `measure` (vegas/montecarlo.jl:156-161, mcmc/montecarlo.jl:166-169) with a nested observable
    shape obs = [0.0, [0.0, 0.0]] (test/montecarlo.jl:53-92): the same source text runs on the GPU and, gcc-compiled,
    in the oracle."""
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.sphere2(), measure=mci.Measure(MEASURE_BODY))
    assert eng.nobs == 3
    ocfg = oracle.Config([ocont()], [[2], [3]], obs_nbin=[1, 2])
    ocfg.set_measure(oracle.compile_c_measure(MEASURE_BODY))
    osolver = dict(vegas=oracle.VEGAS, vegasmc=oracle.VEGASMC, mcmc=oracle.MCMC)[solver]
    got = eng.iteration(solver, 4000, 0, 4, iteration=1, seed=SEED, nchain=8)
    ref = ocfg.iteration(osolver, "sphere2", None, 4000, 0, 4, 1, SEED, nchain=8)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    assert got[2] == pytest.approx(2.0 * got[1], rel=1e-12)   # obs[2][2] accumulates twice obs[2][1]


def test_user_snippet_matches_gcc_compiled_oracle(oracle):
    """an arbitrary user integrand: the same C text JIT-compiled for gfx950 and gcc-compiled for the oracle."""
    body = "w[0] = exp(-x[0]) * cos(3.0 * x[1]) + ud[0] * x[2] * x[2];"
    cfg = mci.Configuration(var=mci.Continuous(0.0, 2.0), dof=[[3]], seed=SEED)
    eng = mci.Engine(cfg, mci.Integrand(body, [0.25]))
    ocfg = oracle.Config([ocont(0, 0.0, 2.0)], [[3]])
    fn = oracle.compile_c_integrand(body)
    got = eng.iteration("vegas", 5000, 0, 4, iteration=0, seed=SEED)
    ref = ocfg.iteration(oracle.VEGAS, fn, [0.25], 5000, 0, 4, 0, SEED)
    np.testing.assert_allclose(got, ref, rtol=1e-9)


def test_launch_geometry_independence(oracle):
    """size-independent property: the RNG stream is keyed by the sample index, so any (threads,
    workgroups-per-block) decomposition gives the same sums up to reassociation."""
    L = math.sqrt(50.0)
    outs = []
    for threads, wpb in ((256, 0), (64, 3), (512, 1), (1024, 7)):
        cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=SEED)
        eng = mci.Engine(cfg, mci.catalog.gaussian(16), threads=threads, wg_per_block=wpb)
        outs.append(eng.iteration("vegas", 20000, 0, 4, iteration=0, seed=SEED))
    for o in outs[1:]:
        np.testing.assert_allclose(o, outs[0], rtol=1e-10)


def test_table_modes_agree(overrides):
    """LDS-resident tables (mode 0), LDS grids + global f64 atomics (1), everything from L2 (2)."""
    outs = []
    for mode, tile_bins, keep_tile0 in (("0", None, "0"), ("1", None, "0"), ("2", None, "0"), ("3", None, "0"), ("3", "1000", "0"), ("3", "2000", "0"),
                                        ("3", "1000", "1")):
        overrides.set("table_mode", int(mode))
        overrides.set("no_split_all", int(keep_tile0))       # "1": tile 0 stays in the sample pass, only the other tiles are replayed
        if tile_bins:
            overrides.set("hist_tile_bins", int(tile_bins))  # force 3 / 2 histogram tiles (split-all: every tile replayed, edges cached in LDS)
        cfg = mci.Configuration(var=mci.Continuous([(0.0, PI)] * 3), dof=[[1]], seed=SEED)
        eng = mci.Engine(cfg, mci.catalog.singular2())
        assert eng.table_mode == int(mode)
        outs.append(eng.iteration("vegas", 8000, 0, 4, iteration=0, seed=SEED))
        eng.run("vegas", 8000, 0, 4, 1, SEED)  # second launch: global histogram was reset by the merge
        second = eng.get_packed()
        assert np.all(np.isfinite(second))
    for o in outs[1:]:
        np.testing.assert_allclose(o, outs[0], rtol=1e-10)


@pytest.mark.parametrize("solver", ["vegasmc", "mcmc"])
def test_chain_solvers_with_tiled_histograms_match_oracle(oracle, solver, overrides):
    """table mode 3 with several histogram tiles under the chain solvers: every tile's workgroup replays the same
    chains (same Philox indices) and keeps its own tile; the merged histogram must equal the untiled oracle's."""
    overrides.set("table_mode", 3)
    overrides.set("hist_tile_bins", 1000)   # one 999-bin leaf per tile -> 3 tiles
    cfg = mci.Configuration(var=mci.Continuous([(0.0, PI)] * 3), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.singular2())
    assert eng.table_mode == 3
    ocfg = oracle.Config([ocont(0, 0.0, PI) for _ in range(3)], [[1]])
    osolver = dict(vegasmc=oracle.VEGASMC, mcmc=oracle.MCMC)[solver]
    got = eng.iteration(solver, 3200, 0, 4, iteration=0, seed=SEED, nchain=16)
    ref = ocfg.iteration(osolver, "singular2", None, 3200, 0, 4, 0, SEED, nchain=16)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    # a host measure closure next to tiled histograms: the records come from the tile-0 workgroups only, once per chain
    seen = []
    def m(x, obs, weights, config):
        seen.append(len(weights[0]))
        obs[0][0] += weights[0].sum()
    cfg2 = mci.Configuration(var=mci.Continuous([(0.0, PI)] * 3), dof=[[1]], seed=SEED)
    eng2 = mci.Engine(cfg2, mci.catalog.singular2(), measure=m)
    got2 = eng2.iteration(solver, 3200, 0, 4, iteration=0, seed=SEED, nchain=16)
    np.testing.assert_allclose(got2, ref, rtol=1e-9, atol=1e-300)
    assert len(seen) == 4 and all(16 <= n <= 3200 + 16 for n in seen)   # (:mcmc measures at i = nburnin .. neval + nburnin inclusive, mcmc/montecarlo.jl:134,143)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
def test_degenerate_pools_match_oracle(oracle, solver):
    """ragged layouts the updates special-case: a single-valued Discrete (nothing to sample: vegas_mc/updates.jl:52-54,
    mcmc/updates.jl:79-81), a pool no integrand uses (maxdof = 0: vegas_mc/updates.jl:55-57, mcmc/updates.jl:82) and
    integrands with different dof on several pools."""
    var = (mci.Continuous(0.0, 2.0), mci.Discrete(3, 3), mci.Continuous(0.0, 1.0), mci.Discrete(1, 4))
    dof = [[2, 1, 0, 1], [1, 1, 0, 0]]
    body = "w[0] = x[0] * x[1] * x[2] * x[3]; w[1] = exp(-x[0]) * x[2];"   # draws: X1, X2, D3, (pool 2 unused), D4
    cfg = mci.Configuration(var=var, dof=dof, seed=SEED)
    eng = mci.Engine(cfg, mci.Integrand(body))
    assert eng.ndraw == 4
    ocfg = oracle.Config([ocont(0, 0.0, 2.0), odisc(1, 3, 3), ocont(2, 0.0, 1.0), odisc(3, 1, 4)], dof)
    fn = oracle.compile_c_integrand(body)
    osolver = dict(vegas=oracle.VEGAS, vegasmc=oracle.VEGASMC, mcmc=oracle.MCMC)[solver]
    got = eng.iteration(solver, 3200, 0, 4, iteration=0, seed=SEED, nchain=8)
    ref = ocfg.iteration(osolver, fn, None, 3200, 0, 4, 0, SEED, nchain=8)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    if solver == "mcmc":
        # the single-valued Discrete never moves; it must not count as an endless hold (it would push the automatic
        # chain length to one chain per block): were it counted, every chain's longest hold would be its full length
        # (400 measured + 496 burn-in steps -> bucket 10)
        hh = eng.hold_histogram()
        np.testing.assert_array_equal(hh, ocfg.hold_hist)
        assert hh.sum() == 4 * 8 and hh[10] < hh.sum() and hh[11:].sum() == 0, hh
    r = eng.integrate(solver, neval=64000, niter=5, block=8, seed=SEED)
    # exact: int_0^2 x dx * int_0^2 y dy * 3 * sum_{1..4} d = 2*2*3*10 = 120 ; int_0^2 e^-x dx * 3 = 3 (1 - e^-2)
    exact = np.array([120.0, 3.0 * (1.0 - math.exp(-2.0))])
    assert np.all(np.abs(r["mean"] - exact) < 7.0 * r["stdev"]), (r["mean"], r["stdev"])


@pytest.mark.parametrize("nchain", [1, 16])
def test_mcmc_with_a_fermik_variable_matches_oracle(oracle, nchain):
    """FermiK{3} (variable.jl:1-20, sampler.jl:109-281): joint create!/remove!/shift!/swap! of a 3-component slot under
    :mcmc, two integrands with 1 and 2 momenta so that changeIntegrand creates/removes slots and swapVariable has work."""
    kF, beta = 1.9191582926775128, 6.787997763336297
    var = (mci.Continuous(0.0, beta, alpha=3.0), mci.FermiK(3, kF, 0.2 * kF, 10.0 * kF), mci.Discrete(1, 4, adapt=False))
    dof = [[1, 1, 1], [1, 2, 1]]
    body = """
    const double k2a = x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    w[0] = exp(-0.3 * x[0]) * exp(-0.5 * k2a) * (1.0 + 0.1 * x[7]);
    if (idx == 0) return;                                   // an :mcmc chain on integrand 1 has no second momentum
    const double k2b = x[4] * x[4] + x[5] * x[5] + x[6] * x[6];
    w[1] = exp(-0.2 * x[0]) * exp(-0.5 * (k2a + k2b)) * (1.0 + 0.05 * x[7] + 0.1 * x[1] * x[4]);"""
    cfg = mci.Configuration(var=var, dof=dof, seed=SEED)
    eng = mci.Engine(cfg, mci.Integrand(body))
    assert eng.ndraw == 1 + 2 * 3 + 1
    ocfg = oracle.Config([ocont(0, 0.0, beta, alpha=3.0), dict(kind=2, pool=1, lower=kF, upper=0.2 * kF, npts=3, alpha=10.0 * kF),
                          odisc(2, 1, 4, adapt=False)], dof)
    obody = body.replace("if (idx == 0) return;", "")      # the oracle evaluates every output (unused ones are ignored)
    fn = oracle.compile_c_integrand(obody)
    got = eng.iteration("mcmc", 3200, 0, 4, iteration=3, seed=SEED, nchain=nchain)
    ref = ocfg.iteration(oracle.MCMC, fn, None, 3200, 0, 4, 3, SEED, nchain=nchain)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    np.testing.assert_array_equal(eng.hold_histogram(), ocfg.hold_hist)
    with pytest.raises(mci.MCIError):                       # "vegas doesn't work with FermiK variable yet"  test/bubble_FermiK.jl:2
        eng.iteration("vegas", 3200, 0, 4, iteration=0, seed=SEED)


def test_error_paths():
    """non-positive normalization (main.jl:269-271) and non-finite histogram (variable.jl:212) surface as errors."""
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.Integrand("w[0] = 1.0 / (x[0] - x[0]);"))  # inf everywhere
    eng.run("vegas", 1000, 0, 2, 0, SEED)
    with pytest.raises(mci.MCIError) as e:
        eng.finish("vegas", 2, adapt=True)
    assert e.value.code == 5 and "finite" in str(e.value)


def test_chain_streams_are_addressed_by_block_and_chain(oracle):
    """A chain is (block, chain within the block) -- DESIGN.md section 3: its draws do not depend on how many chains any
    other block runs, so blocks run with DIFFERENT chain counts (what every rank's own :mcmc measurement leads to) never
    share a stream, and a block computed alone equals the same block computed next to others."""
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    npb = 3200
    both = eng.iteration("mcmc", npb, 0, 2, iteration=1, seed=SEED, nchain=8)
    b0 = eng.iteration("mcmc", npb, 0, 1, iteration=1, seed=SEED, nchain=8)
    b1 = eng.iteration("mcmc", npb, 1, 2, iteration=1, seed=SEED, nchain=8)
    n = eng.nobs
    off = 1e-10 * 2                                     # every partial result carries the clearStatistics offsets once more
    np.testing.assert_allclose(b0[:2 * n] + b1[:2 * n], both[:2 * n], rtol=1e-12)
    np.testing.assert_allclose(b0[2 * n + 2:] + b1[2 * n + 2:], both[2 * n + 2:], rtol=1e-9, atol=1e-6)
    # under the old addressing (g = block*nchain + ch) block 1 with 4 chains replayed chains 4..7 of block 0 with 8 chains
    o8 = ocfg.iteration(oracle.MCMC, c["oname"], c["ud"], npb, 0, 1, 1, SEED, nchain=8)
    o4 = ocfg.iteration(oracle.MCMC, c["oname"], c["ud"], npb, 1, 2, 1, SEED, nchain=4)
    g4 = eng.iteration("mcmc", npb, 1, 2, iteration=1, seed=SEED, nchain=4)
    np.testing.assert_allclose(g4[:2 * n], o4[:2 * n], rtol=1e-9)
    assert not np.allclose(o4[:n], o8[:n], rtol=1e-3)
    with pytest.raises(mci.MCIError):                   # the block index has 12 bits of the stream word
        eng.iteration("mcmc", npb, 4095, 4097, iteration=1, seed=SEED, nchain=4)


@pytest.mark.parametrize("threads,phase", [(None, None), (768, None), (512, None), (1024, "0")], ids=["plan_a_1024", "768", "plan_b_512", "no_phase_1024"])
def test_c4_genz32_runs_in_l2_table_mode_and_matches_oracle(oracle, threads, phase, overrides):
    """BASELINE config 4 layout: 32 independent grids (256 KB of edges > LDS).  Default: one workgroup per CU walking the gathered grids
    dimension-major, the largest of 1024 / 768 / 512 threads at which the sample pass shows no scratch (plan A: 1024 for the Genz
    integrand now that the code object holds ONE loop variant, 120 VGPRs); 512 threads is plan B, the fallback for integrands that
    need more than 168 registers; the override l1_phase = 0 draws in the natural order."""
    if phase is not None:
        overrides.set("l1_phase", int(phase))
    ud = genz_userdata(32)
    cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.genz_product_peak(32), **(dict(threads=threads) if threads else {}))
    assert eng.table_mode == 3  # histograms in LDS (2 tiles of 16 grids), edges gathered from L2
    ocfg = oracle.Config([ocont(0) for _ in range(32)], [[1]])
    got = eng.iteration("vegas", 2000, 0, 4, iteration=0, seed=SEED)
    ref = ocfg.iteration(oracle.VEGAS, "genz_product_peak", ud, 2000, 0, 4, 0, SEED)
    np.testing.assert_allclose(got[:4], ref[:4], rtol=1e-11)
    np.testing.assert_allclose(got[4:], ref[4:], rtol=1e-9)
    assert eng.kernel_times_ms(1)[2] == (threads or 1024)
    assert genz_exact(32) > 0


def test_mid_size_launches_of_light_integrands_use_wide_workgroups_and_the_timed_launches_report_their_clock(oracle):
    """(i) A plain-layout :vegas kernel of a light integrand (at most 8 draws, <= 128 registers) is compiled with a launch bound of 512
    threads and its mid-size launches -- 2^19 <= samples x draws, samples < 2^22: the sizes of the reference's own tests
    (test/montecarlo.jl:298-387) -- run 256 workgroups of 512 threads (profiles/r05_latency.txt); bigger and smaller launches keep the
    256-thread geometry; same sums whatever the geometry.  (ii) Every timed :vegas launch brackets its sample loop with s_memtime /
    s_memrealtime (mci_kernel_clocks): a plausible shader clock comes back -- what bench.py prices data-sheet issue cycles with."""
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    eng.set_kernel_timing(1)
    block = 16
    for npb, want in ((62500, 512), (1000, 256), (1 << 19, 256)):
        got = eng.iteration("vegas", npb, 0, block, iteration=0, seed=SEED)
        assert eng.kernel_times_ms(1)[2] == want, (npb, eng.kernel_times_ms(1))
        if npb <= 62500:
            ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
            gs, gh = hist_split(got, eng.nobs, cfg.N)
            rs, rh = hist_split(ref, eng.nobs, cfg.N)
            np.testing.assert_allclose(gs, rs, rtol=1e-11, atol=1e-300)
            np.testing.assert_allclose(gh, rh, rtol=1e-9)
    clk = eng.kernel_clocks_mhz(8)
    assert len(clk) >= 3 and np.all((clk > 500.0) & (clk < 3000.0)), clk
    narrow = mci.Engine(mci.Configuration(var=c["var"](), dof=c["dof"], seed=SEED), c["f"], threads=256)   # an explicit size: the kernel follows it
    narrow.set_kernel_timing(1)
    b = narrow.iteration("vegas", 62500, 0, block, iteration=0, seed=SEED)
    assert narrow.kernel_times_ms(1)[2] == 256
    a = eng.iteration("vegas", 62500, 0, block, iteration=0, seed=SEED)
    np.testing.assert_allclose(a, b, rtol=1e-10)
