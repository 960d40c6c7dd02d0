"""GPU tests of the persistent :vegas launch (mci_set_persistent; csrc/mci_train.h vegas_persist): all iterations of a launch-bound
integrate() call over ONE Continuous variable type as ONE launch -- sample -> grid-wide arrive -> every workgroup refines its own copy
of the map, a statistics workgroup merges the blocks -> next iteration -- against the oracle's loop (main.jl:142-207) and against the
launch chain on the same Philox streams.

Tolerances are those of the launch chain's own run-level tests (test_hip_parity.test_full_integrate_matches_oracle[prefix]): the
refinement is the prefix-scan walk, whose whole-run agreement with the reference recurrence is 1e-4 (< 0.05 sigma); a single iteration
(no train! in between) agrees to 1e-11."""
import numpy as np
import pytest

import mcintegration_jl_amd as mci
from layout_cases import check_persistent_call
from test_hip_parity import CASES, SEED, make

pytestmark = pytest.mark.gpu

NAMES = ["c1_log_over_sqrt", "sphere2_padding", "hypersphere", "c2_gauss16_shared_pool", "c5_nested_gauss"]   # one Continuous leaf
OTHERS = ["bubble", "c2_gauss4_composite", "discrete", "discrete2_composite", "singular2_composite"]           # several leaves / Discrete


@pytest.mark.parametrize("name", NAMES)
def test_persistent_run_matches_oracle(oracle, name):
    """the whole loop in one launch vs the oracle's loop, same seed: every iteration's mean / error, the final Result, the trained maps"""
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_persistent("on")
    r = eng.integrate("vegas", neval=40000, niter=6, block=16, seed=SEED)
    assert eng.last_integrate_persistent()
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=40000, niter=6, block=16, seed=SEED)
    # first iteration: nothing but reassociation between the two
    np.testing.assert_allclose(r["iter_mean"][0], o["iter_mean"][0], rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(r["iter_std"][0], o["iter_std"][0], rtol=1e-8, atol=1e-300)
    rtol = 1e-4
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=rtol, atol=1e-300)
    np.testing.assert_allclose(r["iter_std"], o["iter_std"], rtol=100 * rtol, atol=1e-300)
    np.testing.assert_allclose(r["mean"], o["mean"], rtol=rtol)
    np.testing.assert_allclose(r["stdev"], o["stdev"], rtol=100 * rtol)
    np.testing.assert_allclose(r["chi2"], o["chi2"], rtol=1000 * rtol, atol=1e-9)
    assert np.all(np.abs(r["mean"] - o["mean"]) < 5e-2 * o["stdev"])
    assert r["neval"] == 6 * 40000
    for i, lf in enumerate(c["oleaves"]):
        if lf["kind"] == 0:
            g, og = eng.grid(i), ocfg.grid(i)
            assert g[0] == og[0] and g[-1] == og[-1] and np.all(np.diff(g) > 0)
            np.testing.assert_allclose(g, og, rtol=0, atol=1e-4 * (lf["upper"] - lf["lower"]))
        elif lf.get("adapt", True):
            np.testing.assert_allclose(eng.distribution(i)[0], ocfg.distribution(i), rtol=1e-4)


@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "c5_nested_gauss", "c2_gauss16_shared_pool"])
def test_one_persistent_iteration_leaves_the_oracles_packed_buffer_and_grid(oracle, name):
    """niter = 1: the packed buffer the launch leaves behind (statistics head, merged histograms cleared by train!) and the map
    after ONE train! -- the per-iteration tolerances of the launch chain (test_train_matches_oracle)"""
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_persistent("on")
    block, npb = 8, 4000
    r = eng.integrate("vegas", neval=block * npb, niter=1, block=block, seed=SEED, ignore=0)
    assert eng.last_integrate_persistent()
    packed = ocfg.iteration(oracle.VEGAS, c["oname"], c["ud"], npb, 0, block, 0, SEED)
    ocfg.train()
    om, oe = oracle.mean_std(packed[:eng.nobs], packed[eng.nobs:2 * eng.nobs], block)
    np.testing.assert_allclose(r["iter_mean"][0], om, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(r["iter_std"][0], oe, rtol=1e-8, atol=1e-300)
    nstat = 2 * eng.nobs + 2 + cfg.N + 1
    np.testing.assert_allclose(eng.iteration_log(1)[0], packed[:nstat], rtol=1e-11, atol=1e-300)
    got = eng.get_packed()
    np.testing.assert_allclose(got[:nstat], packed[:nstat], rtol=1e-11, atol=1e-300)
    for i, lf in enumerate(c["oleaves"]):
        if lf["kind"] == 0:
            g, og = eng.grid(i), ocfg.grid(i)
            assert g[0] == og[0] and g[-1] == og[-1]
            np.testing.assert_allclose(g, og, rtol=0, atol=1e-12 * (lf["upper"] - lf["lower"]))
        elif lf.get("adapt", True):
            d, a = eng.distribution(i)
            np.testing.assert_allclose(d, ocfg.distribution(i), rtol=1e-11)
            np.testing.assert_allclose(a, ocfg.accumulation(i), rtol=1e-11)


@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "sphere2_padding", "c2_gauss16_shared_pool"])
def test_persistent_launch_and_launch_chain_give_the_same_run(name, oracle):
    """the same call with the persistent launch on and off, then a second call that continues from the trained map (resume pattern
    docs/src/index.md:129): the two paths hand over to each other"""
    runs = {}
    for mode in ("on", "off"):
        c, cfg, eng, ocfg = make(name, oracle)
        eng.set_persistent(mode)
        a = eng.integrate("vegas", neval=20000, niter=5, block=16, seed=SEED)
        assert eng.last_integrate_persistent() == (mode == "on")
        eng.set_persistent("off" if mode == "on" else "on")   # hand over to the other path
        b = eng.integrate("vegas", neval=20000, niter=3, block=16, seed=SEED, first_iteration=5)
        assert eng.last_integrate_persistent() == (mode == "off")
        runs[mode] = (a, b, eng.grid(0))
    for k in (0, 1):
        np.testing.assert_allclose(runs["on"][k]["iter_mean"], runs["off"][k]["iter_mean"], rtol=1e-4, atol=1e-300)
        np.testing.assert_allclose(runs["on"][k]["iter_std"], runs["off"][k]["iter_std"], rtol=1e-2, atol=1e-300)
    np.testing.assert_allclose(runs["on"][2], runs["off"][2], rtol=0, atol=1e-4 * (runs["on"][2][-1] - runs["on"][2][0]))


@pytest.mark.parametrize("name", OTHERS)
def test_layouts_without_a_persistent_kernel_take_the_launch_chain(oracle, name):
    """several leaves (every workgroup would have to refine all of them) or a Discrete one: the call runs as the launch chain, whatever
    mci_set_persistent says, with the launch chain's results"""
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_persistent("on")
    r = eng.integrate("vegas", neval=40000, niter=3, block=16, seed=SEED)
    assert not eng.last_integrate_persistent()
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=40000, niter=3, block=16, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-4, atol=1e-300)


def test_persistent_launch_without_adaptation_and_many_iterations(oracle):
    """adapt = false: no train!, the map stays; 40 iterations in one launch (the counters only grow), twice (they carry over)"""
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    eng.set_persistent("on")
    g0 = eng.grid(0).copy()
    r1 = eng.integrate("vegas", neval=16000, niter=40, block=16, seed=SEED, adapt=False, ignore=0)
    r2 = eng.integrate("vegas", neval=16000, niter=40, block=16, seed=SEED, adapt=False, ignore=0, first_iteration=40)
    assert eng.last_integrate_persistent()
    assert np.array_equal(eng.grid(0), g0)
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=16000, niter=80, block=16, seed=SEED, adapt=False, ignore=0)
    np.testing.assert_allclose(np.concatenate([r1["iter_mean"], r2["iter_mean"]]), o["iter_mean"], rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(np.concatenate([r1["iter_std"], r2["iter_std"]]), o["iter_std"], rtol=1e-8, atol=1e-300)


def test_automatic_mode_takes_the_persistent_launch_for_launch_bound_calls_only():
    """integrate() at the reference's default size (neval = 1e4, main.jl:76) goes persistent once its code object exists; a call of
    1e7 samples, a chain solver, measurefreq != 1 and a forced geometry take the launch chain"""
    import os
    import time
    # (a body no kernel cache has seen: the rule below counts the calls of a kernel per process and skips the count when its code object exists)
    src = "return x[0] * x[0] + x[1] * x[1] + 1e-300 * %d;" % (os.getpid() * 1000003 + int(time.time() * 1e3) % 1000003)
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    res = mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
    eng = cfg._engine
    assert not eng.last_integrate_persistent()   # (a new kernel: its persistent form is not in the kernel cache, a few calls do not ask for it)
    for k in range(2000):   # the 256th such call starts the larger translation unit on its own thread; the calls go on as a launch chain
        res = mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
        if eng.last_integrate_persistent():
            break
        if k > 260:
            time.sleep(0.02)
    assert eng.last_integrate_persistent() and k >= 254, k
    assert abs(res.mean[0] - 2.0 / 3.0) < 6 * res.stdev[0]
    mci.integrate(src, config=cfg, solver="vegas", neval=1e7, niter=2)
    assert not eng.last_integrate_persistent()
    mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
    assert eng.last_integrate_persistent()
    mci.integrate(src, config=cfg, solver="vegas", neval=1e4, measurefreq=2)
    assert not eng.last_integrate_persistent()
    mci.integrate(src, config=cfg, solver="vegasmc", neval=1e4)
    assert not eng.last_integrate_persistent()
    eng.set_launch(256, 2)
    mci.integrate(src, config=cfg, solver="vegas", neval=1e4)
    assert not eng.last_integrate_persistent()


@pytest.mark.parametrize("case_id", [3, 17, 41, 72, 106, 108, 150, 211, 260, 301, 333, 399])
def test_random_persistent_calls_match_oracle(oracle, case_id):
    """a dozen cases of the randomised campaign (tools/fuzz_layouts.py --persist, profiles/r03_fuzz_persistent.txt: 400 cases): random grid
    size (17..1500 increments), learning rate, dof tables, 1-40 blocks, 4..123457 samples, 1-6 iterations, with and without adaptation"""
    what, persistent = check_persistent_call(oracle, case_id)
    assert persistent, what
    oracle.set_rng_rounds(10)


def test_two_processes_share_the_device_with_persistent_launches():
    """co-residency: a persistent grid is at most 129 workgroups of <= 64 KiB of LDS, so two of them (two processes on one GPU, the way
    two ranks of a test job share a device) fit the chip together and neither stalls in its grid-wide waits (a stall would fail the call
    with MCI_ERR_HIP after 2 s); both get the single-process numbers"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import mcintegration_jl_amd as mci\n"
            "cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]], seed=11)\n"
            "eng = mci.Engine(cfg, mci.catalog.x2y2()); eng.set_persistent('on')\n"
            "out = []\n"
            "for k in range(300):\n"
            "    r = eng.integrate('vegas', neval=100000, niter=20, block=16, seed=11, first_iteration=20 * k)\n"
            "    assert eng.last_integrate_persistent()\n"
            "    out.append(r['mean'][0])\n"
            "print('%%.17g %%.17g' %% (out[0], out[-1]))\n" % root)
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    a, b = (tuple(float(v) for v in so.split()) for so, _ in outs)
    assert abs(a[0] - 2.0 / 3.0) < 1e-3 and abs(a[1] - 2.0 / 3.0) < 1e-3
    np.testing.assert_allclose(a, b, rtol=1e-6)   # same seeds, same iterations: the same runs up to the order of the histogram atomics


def test_a_stalled_persistent_launch_falls_back_to_the_launch_chain(oracle):
    """The grid-wide wait of the persistent launch needs all its workgroups resident; on a device shared with another long-running kernel
    a wait can run out of time.  The call must not be lost: the map is written back after the LAST turn only, so mci_integrate resets the
    counters and histogram buffers, rewinds the iteration log and runs the SAME iterations through the launch chain (and later calls take
    the chain).  Forced here with the test hook of csrc/mci_debug.h: a wait of one 10 ns tick."""
    from mcintegration_jl_amd._lib import check, lib
    c, cfg, eng, ocfg = make("c1_log_over_sqrt", oracle)
    eng.set_persistent("on")
    check(lib().mci_debug_persist_spin_ticks(eng.p, 1))
    r = eng.integrate("vegas", neval=40000, niter=6, block=16, seed=SEED)
    assert not eng.last_integrate_persistent()                      # it ran as a launch chain in the end
    c2, cfg2, chain, _ = make("c1_log_over_sqrt", oracle)
    chain.set_persistent("off")
    q = chain.integrate("vegas", neval=40000, niter=6, block=16, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], q["iter_mean"], rtol=1e-9)      # the same iterations as a chain that never tried
    np.testing.assert_allclose(r["iter_std"], q["iter_std"], rtol=1e-6)
    np.testing.assert_allclose(eng.grid(0), chain.grid(0), rtol=0, atol=1e-9)
    o = ocfg.integrate(oracle.VEGAS, c["oname"], c["ud"], neval=40000, niter=6, block=16, seed=SEED)
    np.testing.assert_allclose(r["iter_mean"], o["iter_mean"], rtol=1e-4)
    assert r["neval"] == 6 * 40000
    check(lib().mci_debug_persist_spin_ticks(eng.p, 200000000))
    r2 = eng.integrate("vegas", neval=40000, niter=3, block=16, seed=SEED, first_iteration=6, ignore=0)     # later calls: the launch chain
    q2 = chain.integrate("vegas", neval=40000, niter=3, block=16, seed=SEED, first_iteration=6, ignore=0)
    assert not eng.last_integrate_persistent()
    np.testing.assert_allclose(r2["iter_mean"], q2["iter_mean"], rtol=1e-9)
    eng.check_status()
