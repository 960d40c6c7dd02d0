"""GPU parity tests of the STEADY-STATE trips of the sample-batch kernels against the CPU oracle.

The launch rule gives a small launch one sample per lane (`wg_per_block <= ceil(neval_per_block / threads)`), so the oracle
comparisons of tests/test_hip_parity.py only ever meet the first trip / the tail of a kernel's sample loop.  Here the geometry is
forced (`wg_per_block` = 1 or 2) and a block is 5e4 .. 2e5 samples long, so that every lane runs tens to hundreds of trips of
the loop that the full-size launches spend their time in -- the two-samples-per-trip software pipeline of the headline kernel with
its ping-pong pending records (mci_device.h vegas_batch / draw_sample_pipe), the phased gather trips and the 8-samples-per-lane
replay of the many-grid plans (draw_gather_phase_pipe, vegas_tiles), the carried measurement remainder (measurefreq > 1), several
chains per lane -- and the packed result is compared with the oracle's on the same Philox streams
(reference loop: src/vegas/montecarlo.jl:117-187; chains: src/vegas_mc/montecarlo.jl:184-232, src/mcmc/montecarlo.jl:133-172).

Tolerances as in test_hip_parity.py: statistics head rel 1e-11 (:vegas; sums of up to 2e5 fp64 terms, reassociation only),
histograms rel 1e-9 per bin, counters exact.
"""
import math

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from catalog_params import bubble_userdata, genz_userdata
from layout_cases import check_carried_iterations, pipe_case

pytestmark = pytest.mark.gpu
PI = math.pi
SEED = 20240229
L = math.sqrt(50.0)


def ocont(pool=0, lo=0.0, hi=1.0, **kw):
    return dict(kind=0, pool=pool, lower=lo, upper=hi, **kw)


def odisc(pool, lo, hi, **kw):
    return dict(kind=1, pool=pool, lower=lo, upper=hi, **kw)


def split(packed, nobs, ni):
    n = 2 * nobs + 2 + ni + 1
    return packed[:n], packed[n:]


def compare(got, ref, nobs, ni, rtol_stat=1e-11, rtol_hist=1e-9):
    gs, gh = split(got, nobs, ni)
    rs, rh = split(ref, nobs, ni)
    np.testing.assert_allclose(gs, rs, rtol=rtol_stat, atol=1e-300)
    np.testing.assert_allclose(gh, rh, rtol=rtol_hist)


HEADLINE = {
    # BASELINE configs[1] (C2) and the :vegas form of configs[4] (C5): the kernels of pipe_eligible()
    "c2": dict(var=lambda: mci.Continuous(-L, L), dof=[[16]], f=lambda: mci.catalog.gaussian(16), oleaves=[ocont(0, -L, L)],
               oname="gaussian", ud=[16.0]),
    "c5": dict(var=lambda: mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], f=lambda: mci.catalog.nested_gauss(), oleaves=[ocont()],
               oname="nested_gauss", ud=[4.0, 3.0, 6.0, 9.0, 12.0]),
}


# (case, wg_per_block, rng bits, Philox rounds, measurefreq, samples per block)
#   npb is never a multiple of the launch stride (threads * wg_per_block): lanes below the remainder run one sample more than the
#   others, so odd and even trip counts -- the tail and the no-tail exit of the two-samples-per-trip loop -- meet in every launch
STEADY = [
    ("c2", 1, 52, 10, 1, 200003),   # the bench line's code object (8 copies, 512 threads, VGPR round keys): 390 / 391 samples per lane
    ("c2", 2, 52, 10, 1, 199999),   # 195 / 196 samples per lane
    ("c2", 1, 52, 10, 3, 100003),   # the carried measurement remainder inside the pipelined loop (MF1 = false)
    ("c2", 1, 32, 10, 1, 200003),   # four draws per Philox block: four reads and four atomics per stage
    ("c2", 1, 52, 7, 1, 200003),    # Philox4x32-7
    ("c2", 1, 32, 7, 1, 200003),    # both opt-ins: sixteen copies in one 1024-thread workgroup
    ("c2", 2, 32, 7, 3, 150001),
    ("c5", 1, 52, 10, 1, 200003),   # four integrands with nested dof: per-integrand Jacobians and summed histogram weights in the pending record
    ("c5", 2, 52, 10, 3, 100003),
    ("c5", 1, 32, 10, 1, 100003),
    ("c5", 1, 52, 7, 1, 100003),
]


@pytest.mark.parametrize("name,wpb,bits,rounds,mfreq,npb", STEADY, ids=["%s-wpb%d-%dbit-%dr-mf%d" % t[:5] for t in STEADY])
def test_pipelined_vegas_loop_many_trips_matches_oracle(oracle, name, wpb, bits, rounds, mfreq, npb):
    """The two-samples-per-trip body of the software-pipelined :vegas loop (mci_device.h vegas_batch: `for (; n + stride <
    neval_per_block; n += 2 * stride)`, records pa / pb swapping roles, a sample's atomics landing one trip late, the flush of a lane's
    last sample) under an oracle assertion: vegas/montecarlo.jl:117-187 sample for sample on the same Philox stream."""
    c = HEADLINE[name]
    block = 2
    cfg = mci.Configuration(var=c["var"](), dof=c["dof"], seed=SEED)
    eng = mci.Engine(cfg, c["f"](), wg_per_block=wpb, **({"rng_bits": bits} if bits != 52 else {}), **({"rng_rounds": rounds} if rounds != 10 else {}))
    ocfg = oracle.Config(c["oleaves"], c["dof"])
    if bits == 32:
        ocfg.set_rng_bits(32)
    oracle.set_rng_rounds(rounds)
    try:
        got = eng.iteration("vegas", npb, 1, 1 + block, iteration=3, seed=SEED, measurefreq=mfreq)
        ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 1, 1 + block, 3, SEED, measurefreq=mfreq, nthreads=2)
    finally:
        oracle.set_rng_rounds(10)
    _, wg, threads = eng.kernel_times_ms(1)
    assert wg == block * wpb, (wg, threads)
    if name == "c2":                                        # the copy plan, i.e. the kernel the full-size launches run
        assert threads == (1024 if (bits == 32 and rounds == 7) else 512) and eng.histogram_copies() == (16 if threads == 1024 else 8)
    assert npb // (threads * wpb) >= 70                      # every lane runs >= 35 two-sample trips
    compare(got, ref, eng.nobs, cfg.N)
    n = eng.nobs
    assert got[2 * n + 1] == block * npb                    # config.neval
    measured = sum(((npb // mfreq),) * block)
    assert got[2 * n] == pytest.approx(measured + (block + 1) * 1e-10, rel=1e-13)


def test_pipelined_loop_trained_grid_many_trips_matches_oracle(oracle):
    """the same loop on an ADAPTED map (narrow increments around the peak: the table reads and the histogram adds of a wave pile up
    on a few bank pairs) -- eight training iterations of the oracle, then one long forced-geometry iteration"""
    c = HEADLINE["c2"]
    cfg = mci.Configuration(var=c["var"](), dof=c["dof"], seed=SEED)
    eng = mci.Engine(cfg, c["f"](), wg_per_block=1)
    ocfg = oracle.Config(c["oleaves"], c["dof"])
    ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=800000, niter=8, block=16, seed=SEED, nthreads=8)
    inc = np.diff(ocfg.grid(0))
    assert inc.min() < 0.3 * (2 * L / 999) and inc.max() > 3.0 * (2 * L / 999)     # the map has adapted
    eng.set_grid(0, ocfg.grid(0))
    npb = 150001
    got = eng.iteration("vegas", npb, 0, 2, iteration=7, seed=SEED)
    ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, 2, 7, SEED, nthreads=2)
    compare(got, ref, eng.nobs, cfg.N)


@pytest.mark.parametrize("case_id", range(12))
def test_random_pipelined_layouts_with_forced_geometry_match_oracle(oracle, case_id):
    """a dozen layouts of tools/fuzz_layouts.py --pipe (1-3 Continuous pools, 8..16 draws, ragged dof tables, adapt on / off, 1-4
    integrands, both stream widths, 10 / 7 rounds, measurefreq 1 / 3) with one or two workgroups per block and blocks long enough for
    40..200 samples per lane"""
    rng = np.random.default_rng(7000 + case_id)
    var, oleaves, dof, body, ndraw = pipe_case(rng)
    rounds = int(rng.choice([10, 10, 7]))
    bits = int(rng.choice([52, 52, 32]))
    mfreq = int(rng.choice([1, 1, 3]))
    wpb = int(rng.choice([1, 2]))
    npb = int(rng.choice([20001, 30011, 51234, 77777]))
    it = int(rng.integers(0, 50))
    what = "case %d: pools=%d ni=%d ndraw=%d rounds=%d bits=%d wpb=%d npb=%d measurefreq=%d dof=%s" % (
        case_id, len(var), len(dof), ndraw, rounds, bits, wpb, npb, mfreq, dof)
    cfg = mci.Configuration(var=var, dof=dof, seed=SEED)
    eng = mci.Engine(cfg, mci.Integrand(body), wg_per_block=wpb, rng_rounds=rounds, **({"rng_bits": 32} if bits == 32 else {}))
    assert eng.ndraw == ndraw, what
    fn = oracle.compile_c_integrand(body)
    ocfg = oracle.Config(oleaves, dof)
    if bits == 32:
        ocfg.set_rng_bits(32)
    oracle.set_rng_rounds(rounds)
    try:
        got = eng.iteration("vegas", npb, 0, 2, iteration=it, seed=SEED, measurefreq=mfreq)
        ref = ocfg.iteration(oracle.VEGAS, fn, None, npb, 0, 2, it, SEED, measurefreq=mfreq, nthreads=2)
    finally:
        oracle.set_rng_rounds(10)
    _, wg, threads = eng.kernel_times_ms(1)
    assert wg == 2 * wpb and npb // (threads * wpb) >= 9, (what, wg, threads)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300, err_msg=what + "\n" + body)


@pytest.mark.parametrize("threads,mfreq", [(None, 1), (None, 3), (768, 1), (512, 1)], ids=["plan_a_1024", "plan_a_mf3", "768", "plan_b_512"])
def test_c4_many_grids_many_trips_matches_oracle(oracle, threads, mfreq):
    """BASELINE configs[3] (32 independent grids): ONE workgroup per block walks 48 (1024 threads) / 65 (768) / 97 (512) phased trips of
    the split-all sample pass -- hand-pipelined gather phase, LDS edge cache, bins packed as drawn, weights and bins parked in HBM --
    and the replay kernel 7 / 9 / 13 trips of 8 (4) samples per lane into its two skewed bin-major tiles.  (measurefreq = 3 runs the
    any-cadence code object, whose loop carries the remainder: it may sit one rung lower on the 1024 / 768 / 512 ladder.)"""
    ud = genz_userdata(32)
    cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.genz_product_peak(32), wg_per_block=1, **(dict(threads=threads) if threads else {}))
    assert eng.table_mode == 3
    ocfg = oracle.Config([ocont(0) for _ in range(32)], [[1]])
    npb = 50001
    got = eng.iteration("vegas", npb, 2, 4, iteration=1, seed=SEED, measurefreq=mfreq)
    ref = ocfg.iteration(oracle.VEGAS, "genz_product_peak", ud, npb, 2, 4, 1, SEED, measurefreq=mfreq, nthreads=2)
    _, wg, th = eng.kernel_times_ms(1)
    assert wg == 2 and (th == threads if threads else th in (1024, 768)), (wg, th)
    compare(got, ref, eng.nobs, cfg.N)
    # equal totals in every grid's histogram (each sample adds its weight once per grid): a bin unpacked from the wrong field would
    # still conserve the total, a dropped or doubled sample would not
    h = split(got, eng.nobs, cfg.N)[1][:32 * 999].reshape(32, 999)
    np.testing.assert_allclose(h.sum(axis=1), h.sum(axis=1)[0], rtol=1e-10)


def test_c2_sixteen_grids_many_trips_matches_oracle(oracle):
    """the 16-independent-grid layout of configs[1]: histogram in the pass (one tile), three grids' edges in the LDS edge
    cache, thirteen gathered from L2, one 1024-thread workgroup -- 97 samples per lane"""
    cfg = mci.Configuration(var=mci.Continuous([(-L, L)] * 16), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.gaussian(16), wg_per_block=1)
    assert eng.table_mode == 3
    ocfg = oracle.Config([ocont(0, -L, L) for _ in range(16)], [[1]])
    npb = 100003
    got = eng.iteration("vegas", npb, 0, 2, iteration=2, seed=SEED)
    ref = ocfg.iteration(oracle.VEGAS, "gaussian", [16.0], npb, 0, 2, 2, SEED, nthreads=2)
    assert eng.kernel_times_ms(1)[1] == 2
    compare(got, ref, eng.nobs, cfg.N)


def _generic_cases():
    bp = bubble_userdata()
    return {
        "c1_log_over_sqrt": dict(var=lambda: mci.Continuous(0.0, 1.0), dof=[[1]], f=lambda: mci.catalog.log_over_sqrt(), oleaves=[ocont()],
                                 oname="log_over_sqrt", ud=None),
        "sphere2_padding": dict(var=lambda: mci.Continuous(0.0, 1.0), dof=[[2], [3]], f=lambda: mci.catalog.sphere2(), oleaves=[ocont()],
                                oname="sphere2", ud=None),
        "hypersphere": dict(var=lambda: mci.Continuous(-1.0, 1.0), dof=[[2], [3], [4]], f=lambda: mci.catalog.hypersphere(3),
                            oleaves=[ocont(0, -1.0, 1.0)], oname="hypersphere", ud=[3.0]),
        "discrete2_composite": dict(var=lambda: mci.Discrete([(1, 3), (1, 4)]), dof=[[1]], f=lambda: mci.catalog.one(),
                                    oleaves=[odisc(0, 1, 3), odisc(0, 1, 4)], oname="one", ud=None),
        "bubble": dict(var=lambda: (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
                                    mci.Continuous(0.0, bp[1], alpha=3.0), mci.Discrete(1, 4, adapt=False)),
                       dof=[[1, 1, 1, 1, 1]], f=lambda: mci.catalog.bubble(), obs=[np.zeros(4)], measure=lambda: mci.bin_by(4),
                       oleaves=[ocont(0, 0, 1, alpha=3.0), ocont(1, 0, PI, alpha=3.0), ocont(2, 0, 2 * PI, alpha=3.0), ocont(3, 0, bp[1], alpha=3.0),
                                odisc(4, 1, 4, adapt=False)],
                       oname="bubble", ud=bp, obs_nbin=[4], obs_bin_draw=[4]),
    }


GENERIC = _generic_cases()


def _make(name, oracle, **eng_kw):
    c = GENERIC[name]
    cfg = mci.Configuration(var=c["var"](), dof=c["dof"], obs=c.get("obs"), seed=SEED)
    eng = mci.Engine(cfg, c["f"](), measure=c["measure"]() if "measure" in c else None, **eng_kw)
    ocfg = oracle.Config(c["oleaves"], c["dof"], obs_nbin=c.get("obs_nbin"), obs_bin_draw=c.get("obs_bin_draw"))
    return c, cfg, eng, ocfg


@pytest.mark.parametrize("name,mfreq", [("c1_log_over_sqrt", 1), ("sphere2_padding", 3), ("hypersphere", 1), ("discrete2_composite", 1),
                                        ("bubble", 1), ("bubble", 3)])
def test_plain_vegas_loop_many_trips_matches_oracle(oracle, name, mfreq):
    """the loop of the kernels outside the pipelined plan (fewer than 8 draws, Discrete / CompositeVar draws, binned observables,
    padding probabilities): one 256-thread workgroup per block, 390 samples per lane"""
    c, cfg, eng, ocfg = _make(name, oracle, wg_per_block=1)
    npb = 100003
    got = eng.iteration("vegas", npb, 0, 2, iteration=4, seed=SEED, measurefreq=mfreq)
    ref = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, 2, 4, SEED, measurefreq=mfreq, nthreads=2)
    assert eng.kernel_times_ms(1)[1] == 2
    compare(got, ref, eng.nobs, cfg.N)


@pytest.mark.parametrize("solver", ["vegasmc", "mcmc"])
@pytest.mark.parametrize("name", ["sphere2_padding", "bubble"])
def test_several_chains_per_lane_match_oracle(oracle, name, solver):
    """the chain solvers' outer loop (`for ch = slice * T + tid; ch < nchain; ch += wg_per_block * T`): 600 chains of a block on ONE
    256-thread workgroup -- lanes run three or two chains one after the other, their propose / accept counters and holding-time
    records handed over between chains (vegas_mc/montecarlo.jl:184-232, mcmc/montecarlo.jl:133-172 per chain)"""
    c, cfg, eng, ocfg = _make(name, oracle, wg_per_block=1)
    nchain, npb = 600, 600 * 40
    osolver = dict(vegasmc=oracle.VEGASMC, mcmc=oracle.MCMC)[solver]
    kw = {}
    if solver == "mcmc":
        ocfg.set_thermal_ratio(0.1)
        kw = dict(thermal_ratio=0.1)
    got = eng.iteration(solver, npb, 0, 2, iteration=1, seed=SEED, nchain=nchain, measurefreq=2, **kw)
    ref = ocfg.iteration(osolver, c["oname"], c["ud"], npb, 0, 2, 1, SEED, nchain=nchain, measurefreq=2, nthreads=2)
    assert eng.kernel_times_ms(1)[1] == 2
    compare(got, ref, eng.nobs, cfg.N, rtol_stat=1e-9, rtol_hist=1e-8)
    pr, ac = eng.acceptance()
    npa = pr.size
    np.testing.assert_allclose(pr.ravel(), ref[-2 * npa:-npa], rtol=1e-12)
    np.testing.assert_allclose(ac.ravel(), ref[-npa:], rtol=1e-12)
    if solver == "mcmc":
        np.testing.assert_array_equal(eng.hold_histogram(), ocfg.hold_hist)


@pytest.mark.parametrize("solver", ["vegasmc", "mcmc"])
@pytest.mark.parametrize("name", ["sphere2_padding", "bubble", "discrete2_composite"])
def test_carried_chains_match_oracle(oracle, name, solver):
    """Carried chains (mci_set_chain_carry, this engine's many-chain decomposition only): the next iteration of the same solver over
    the same blocks continues the previous launch's chains with the reference's own burn-in only -- :vegasmc chain (block, ch) starts
    from the stored configuration that the block's systematic resampling with probability ~ new target density / old one assigns to
    it (mci_vegasmc_carry_weights + k_resample_chains | mcio_resample_weighted), bins and probabilities looked up again on the map
    train! has just refined, and only out of a launch that ran on a map refined at least once; :mcmc chain (block, ch) from the stored chain (configuration AND integrand index) that the block's systematic
    resampling with probability ~ reweight_new[curr] / reweight_old[curr] assigns to it (k_resample_chains | mcio_resample_chains).
    Four consecutive iterations with doReweight! and train! in between and a chain count that changes (16, 16, 40, 8 per block)
    against the oracle's mirror; a repeated iteration number starts afresh again."""
    c, cfg, eng, ocfg = _make(name, oracle)
    osolver = dict(vegasmc=oracle.VEGASMC, mcmc=oracle.MCMC)[solver]
    block, npb = 4, 4800
    kw = {}
    if solver == "mcmc":
        ocfg.set_thermal_ratio(0.1)
        kw = dict(thermal_ratio=0.1)
    n = eng.nobs
    for it, nch in enumerate([16, 16, 40, 8]):
        got = eng.iteration(solver, npb, 0, block, iteration=it, seed=SEED, nchain=nch, **kw)
        ref = ocfg.iteration(osolver, c["oname"], c["ud"], npb, 0, block, it, SEED, nchain=nch, nthreads=2)
        # (:vegasmc chains are carried only out of a launch that ran on a map train! had refined at least once: the second iteration
        # of a fresh problem starts afresh)
        assert eng.last_chain_launch() == (nch, it > (1 if solver == "vegasmc" else 0)), (it, eng.last_chain_launch())
        compare(got, ref, n, cfg.N, rtol_stat=1e-8, rtol_hist=1e-7)
        if solver == "mcmc":
            np.testing.assert_array_equal(eng.hold_histogram(), ocfg.hold_hist)
        eng.finish(solver, block, adapt=True)                               # doReweight! + train! on the device (main.jl:183-199)
        ocfg.set_reweight(oracle.do_reweight(ocfg.reweight, ref[2 * n + 2: 2 * n + 2 + cfg.N + 1]))
        ocfg.train()
        np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=1e-9)
    # the same iteration number once more: not a continuation -> fresh starts with the many-chain burn-in floors, as before
    got = eng.iteration(solver, npb, 0, block, iteration=3, seed=SEED, nchain=8, **kw)
    assert eng.last_chain_launch() == (8, False)
    eng2 = _make(name, oracle)[2]
    for i in range(len(c["oleaves"])):
        if c["oleaves"][i]["kind"] == 0:
            eng2.set_grid(i, eng.grid(i))
        elif c["oleaves"][i].get("adapt", True):
            eng2.set_distribution(i, eng.distribution(i)[0])
    eng2.set_reweight(eng.reweight())
    eng2.set_chain_carry("off")
    again = eng2.iteration(solver, npb, 0, block, iteration=3, seed=SEED, nchain=8, **kw)
    np.testing.assert_allclose(got, again, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("name", ["sphere2_padding", "bubble"])
def test_vegasmc_chains_go_on_while_the_map_stays_as_it_is(oracle, name):
    """adapt = false (main.jl:192: no train!, doReweight! only): the map a launch's chains ran on is the map of the next launch, so
    :vegasmc chains are carried from the first launch on -- also on a map train! has never refined, where they are NOT carried onto a
    refined one (test_carried_chains_match_oracle).  Three iterations, 16 | 16 | 40 chains per block, against the oracle's mirror."""
    c, cfg, eng, ocfg = _make(name, oracle)
    block, npb = 4, 4800
    n = eng.nobs
    for it, nch in enumerate([16, 16, 40]):
        got = eng.iteration("vegasmc", npb, 0, block, iteration=it, seed=SEED, nchain=nch)
        ref = ocfg.iteration(oracle.VEGASMC, c["oname"], c["ud"], npb, 0, block, it, SEED, nchain=nch, nthreads=2)
        assert eng.last_chain_launch() == (nch, it > 0), (it, eng.last_chain_launch())
        compare(got, ref, n, cfg.N, rtol_stat=1e-8, rtol_hist=1e-7)
        eng.finish("vegasmc", block, adapt=False)                           # doReweight! only (main.jl:183, :192)
        ocfg.set_reweight(oracle.do_reweight(ocfg.reweight, ref[2 * n + 2: 2 * n + 2 + cfg.N + 1]))
        np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=1e-9)


@pytest.mark.parametrize("case_id", [3, 8, 15, 69, 108, 129, 200, 257])
def test_carried_chains_on_random_layouts_match_oracle(oracle, case_id):
    """eight cases of the randomised campaign (tools/fuzz_layouts.py --carry, profiles/r03_fuzz_carry.txt: 300 cases): 1-5 pools, 1-4
    integrands, four consecutive iterations of :vegasmc and of :mcmc with carried chains, chain counts that grow and shrink between
    iterations (the resampling of stored :mcmc chains picks every new chain's ancestor), doReweight! and train! in between"""
    check_carried_iterations(oracle, case_id)
    oracle.set_rng_rounds(10)


@pytest.mark.parametrize("solver", ["vegas", "vegasmc", "mcmc"])
@pytest.mark.parametrize("name", ["c2", "sphere2_padding", "bubble"])
def test_deterministic_mode_is_bit_reproducible_and_on_the_oracle(oracle, name, solver):
    """mci_set_deterministic: a fixed seed gives BIT-IDENTICAL iterations, grids and reweight factors run to run -- what the
    reference's sequential loop under MersenneTwister(seed) gives (configuration.jl:190) -- for every solver, over several iterations
    with train! in between (which would amplify any last-bit difference of a histogram), with many samples per lane and many chains;
    and the deterministic kernels are the same arithmetic: one iteration against the oracle at the usual tolerances."""
    def build():
        if name == "c2":
            c = HEADLINE["c2"]
            cfg = mci.Configuration(var=c["var"](), dof=c["dof"], seed=SEED)
            return c, cfg, mci.Engine(cfg, c["f"](), deterministic=True), oracle.Config(c["oleaves"], c["dof"])
        c, cfg, eng, ocfg = _make(name, oracle, deterministic=True)
        return c, cfg, eng, ocfg
    runs = []
    for rep in range(3):
        c, cfg, eng, ocfg = build()
        r = eng.integrate(solver, neval=400000, niter=5, block=16, seed=SEED)
        grids = [eng.grid(i) for i, lf in enumerate(c["oleaves"]) if lf["kind"] == 0]
        runs.append((r["iter_mean"].copy(), r["iter_std"].copy(), grids, eng.reweight(), eng.get_packed()))
    for other in runs[1:]:
        assert np.array_equal(other[0], runs[0][0]) and np.array_equal(other[1], runs[0][1])
        for g0, g1 in zip(runs[0][2], other[2]):
            assert np.array_equal(g0, g1)
        assert np.array_equal(other[3], runs[0][3]) and np.array_equal(other[4], runs[0][4])
    # ... and the same numbers as the default kernels up to the order of the sums: one iteration against the oracle
    c, cfg, eng, ocfg = build()
    osolver = dict(vegas=oracle.VEGAS, vegasmc=oracle.VEGASMC, mcmc=oracle.MCMC)[solver]
    kw, okw = {}, {}
    if solver == "mcmc":
        ocfg.set_thermal_ratio(0.1)
        kw = dict(thermal_ratio=0.1)
    if solver != "vegas":
        kw["nchain"] = okw["nchain"] = 48
    got = eng.iteration(solver, 20001, 0, 4, iteration=1, seed=SEED, **kw)
    ref = ocfg.iteration(osolver, c["oname"], c["ud"], 20001, 0, 4, 1, SEED, nthreads=2, **okw)
    compare(got, ref, eng.nobs, cfg.N, rtol_stat=1e-9, rtol_hist=1e-8)
    if solver == "vegas":
        assert eng.histogram_copies() == eng.kernel_times_ms(1)[2] // 64    # one copy per wave


def test_deterministic_mode_refuses_layouts_that_need_histogram_tiles():
    cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.genz_product_peak(32), deterministic=True)
    with pytest.raises(mci.MCIError) as e:
        eng.iteration("vegas", 2000, 0, 4, iteration=0, seed=SEED)
    assert "deterministic" in str(e.value)
