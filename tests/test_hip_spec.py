"""GPU parity tests of the chain solvers with SEVERAL LANES PER CHAIN (csrc/mci_spec.h, mci_set_chain_speculation): a group of lanes
steps one chain speculatively along a tree of accept / reject outcomes.  It is the same chain as the oracle's sequential one -- same
streams, same arithmetic per step -- so every case is compared with the oracle at the tolerances of the lane-per-chain kernels
(tests/test_hip_parity.py: block sums 1e-9, histograms 1e-8 / 1e-9, propose / accept tables entry by entry, the :mcmc holding-time
histogram bucket by bucket), for group sizes 2 .. 64 and for trees from the reject chain to the complete binary tree.  The rest of
the GPU suite runs its small chain launches through these kernels as well (automatic group size); lanes = 1 here keeps the
lane-per-chain kernels under test at the same sizes."""
import math

import os

import numpy as np
import pytest

import mcintegration_jl_amd as mci
from test_hip_parity import CASES, SEED, hist_split, make, ocont

pytestmark = pytest.mark.gpu

# (lanes, acceptance the tree is built for, most accept edges): reject chain | binary tree | the solvers' defaults | small groups
TREES = [(1, 0.0, -1), (64, 1e-3, -1), (64, 0.5, -1), (64, 0.35, 2), (16, 0.5, -1), (4, 0.5, -1), (2, 0.5, -1), (8, 0.2, 1), (32, 0.8, 3)]
IDS = ["lane_per_chain", "64_reject_chain", "64_binary", "64_default_mcmc", "16_binary", "4_binary", "2", "8_one_accept", "32_accepting"]


def check_packed(got, ref, eng, cfg, rtol_h):
    gs, gh = hist_split(got, eng.nobs, cfg.N)
    rs, rh = hist_split(ref, eng.nobs, cfg.N)
    np.testing.assert_allclose(gs, rs, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(gh, rh, rtol=rtol_h)
    pr, ac = eng.acceptance()
    nd, m = cfg.N + 1, max(cfg.N + 1, len(cfg.var))
    npa = 3 * nd * m
    np.testing.assert_allclose(pr.ravel(), ref[-2 * npa:-npa], rtol=1e-12)
    np.testing.assert_allclose(ac.ravel(), ref[-npa:], rtol=1e-12)


@pytest.mark.parametrize("tree", TREES, ids=IDS)
@pytest.mark.parametrize("nchain", [1, 5])
@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "sphere2_padding", "bubble", "discrete2_composite", "c5_nested_gauss"])
def test_vegasmc_groups_step_the_oracles_chain(oracle, name, nchain, tree):
    """row a16 with G lanes per chain: nchain = 1 is the reference's chain (vegas_mc/montecarlo.jl:184-232), nchain = 5 leaves groups
    without a chain in the last pass and ragged chain lengths (1283 steps)"""
    lanes, accept, limit = tree
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_chain_speculation(lanes, accept, limit)
    block, npb = 3, 6415
    got = eng.iteration("vegasmc", npb, 1, 1 + block, iteration=1, seed=SEED, nchain=nchain)
    assert eng.last_chain_speculation()[0] == lanes
    ref = ocfg.iteration(oracle.VEGASMC, c["oname"], c["ud"], npb, 1, 1 + block, 1, SEED, nchain=nchain)
    check_packed(got, ref, eng, cfg, 1e-8)


@pytest.mark.parametrize("tree", TREES, ids=IDS)
@pytest.mark.parametrize("nchain", [1, 5])
@pytest.mark.parametrize("name", ["c1_log_over_sqrt", "sphere2_padding", "hypersphere", "bubble", "singular2_composite", "c5_nested_gauss"])
def test_mcmc_groups_step_the_oracles_chain(oracle, name, nchain, tree):
    """row f1 with G lanes per chain (mcmc/montecarlo.jl:134-172, mcmc/updates.jl): sums, histograms, propose / accept, holding times"""
    lanes, accept, limit = tree
    c, cfg, eng, ocfg = make(name, oracle)
    eng.set_chain_speculation(lanes, accept, limit)
    block, npb = 3, 3205
    got = eng.iteration("mcmc", npb, 0, block, iteration=2, seed=SEED, nchain=nchain, thermal_ratio=0.1)
    assert eng.last_chain_speculation()[0] == lanes
    ocfg.set_thermal_ratio(0.1)
    ref = ocfg.iteration(oracle.MCMC, c["oname"], c["ud"], npb, 0, block, 2, SEED, nchain=nchain)
    check_packed(got, ref, eng, cfg, 1e-9)
    hh = eng.hold_histogram()
    np.testing.assert_array_equal(hh, ocfg.hold_hist)
    assert hh.sum() == block * nchain


@pytest.mark.parametrize("tree", [(64, 0.5, -1), (8, 0.3, 2), (64, 1e-3, -1)], ids=["64_binary", "8", "64_reject_chain"])
def test_measurefreq_neighbor_graph_and_burn_in_with_groups(oracle, tree):
    """measurefreq != 1 (montecarlo.jl:213-214, mcmc/montecarlo.jl:144), a custom neighbor graph (configuration.jl:211-221) and the
    burn-in of fresh chains: every lane of a group measures by its OWN step's index"""
    lanes, accept, limit = tree
    cfg = mci.Configuration(var=mci.Continuous(-1.0, 1.0), dof=[[2], [3], [4]], neighbor=[(1, 4), (1, 2), (1, 3), (2, 3)], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.hypersphere(3))
    eng.set_chain_speculation(lanes, accept, limit)
    ocfg = oracle.Config([ocont(0, -1.0, 1.0)], [[2], [3], [4]])
    ocfg.set_neighbor(cfg.neighbor_lists())
    ocfg.set_thermal_ratio(0.25)
    got = eng.iteration("mcmc", 4000, 0, 2, iteration=0, seed=SEED, nchain=4, measurefreq=3, thermal_ratio=0.25)
    ref = ocfg.iteration(oracle.MCMC, "hypersphere", [3.0], 4000, 0, 2, 0, SEED, nchain=4, measurefreq=3)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    eng.set_chain_speculation(lanes, accept, limit)
    got = eng.iteration("vegasmc", 5000, 0, 2, iteration=0, seed=SEED, nchain=3, measurefreq=7)
    ref = ocfg.iteration(oracle.VEGASMC, c["oname"], c["ud"], 5000, 0, 2, 0, SEED, nchain=3, measurefreq=7)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("solver", ["vegasmc", "mcmc"])
@pytest.mark.parametrize("lanes", [64, 4])
def test_whole_runs_of_single_chains_stay_on_the_oracles_trajectory(oracle, solver, lanes):
    """mci_integrate with the reference's one chain per block (the default call's shape): chains, merge, doReweight!, train!, Result"""
    c, cfg, eng, ocfg = make("sphere2_padding", oracle)
    eng.set_chain_speculation(lanes, 0.0, -1)
    r = eng.integrate(solver, neval=32000, niter=4, block=8, seed=SEED, nchain=1)
    assert eng.last_chain_speculation()[0] == lanes
    o = ocfg.integrate(oracle.VEGASMC if solver == "vegasmc" else oracle.MCMC, "sphere2", None, neval=32000, niter=4, block=8, seed=SEED, nchain=1)
    np.testing.assert_allclose(r["iter_mean"][:2], o["iter_mean"][:2], rtol=1e-7)
    assert np.all(np.abs(r["iter_mean"] - o["iter_mean"]) <= 6 * np.hypot(r["iter_std"], o["iter_std"]) + 1e-300)
    np.testing.assert_allclose(eng.reweight(), ocfg.reweight, rtol=1e-3)


@pytest.mark.parametrize("solver", ["vegasmc", "mcmc"])
def test_carried_chains_with_groups_are_the_lane_per_chain_run(solver):
    """chains carried from one iteration to the next (BatchArgs::carry_x; :mcmc: resampled) under the group kernels: the same numbers
    as the lane-per-chain kernels give, iteration by iteration, with the group size changing between iterations"""
    out = []
    for plan in ([1, 1, 1, 1], [16, 4, 64, 2]):
        cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=SEED)
        eng = mci.Engine(cfg, mci.catalog.nested_gauss())
        rows = []
        for it, lanes in enumerate(plan):
            eng.set_chain_speculation(lanes, 0.4, 2)
            got = eng.iteration(solver, 9600, 0, 4, iteration=it, seed=SEED, nchain=12 if it != 2 else 20, thermal_ratio=0.1)
            assert eng.last_chain_speculation()[0] == lanes and eng.last_chain_launch()[1] == (it > (1 if solver == "vegasmc" else 0))
            rows.append(got.copy())
            eng.finish(solver, 4, True, 1.0)
        out.append(rows)
    for a, b in zip(*out):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("solver", ["vegasmc", "mcmc"])
def test_complex_weights_and_user_measure_with_groups(oracle, solver):
    """ComplexF64 weights (test/montecarlo.jl:172-185) and a user `measure` (:71-84) through the group kernels: equal to the
    lane-per-chain kernels' results"""
    res = []
    for lanes in (1, 64, 8):
        cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1], [2]], type=complex, seed=SEED)
        eng = mci.Engine(cfg, mci.Integrand("w[0] = x[0]; w[1] = 0.0; w[2] = 0.5 * x[0]; w[3] = x[0] * x[1];"))
        eng.set_chain_speculation(lanes, 0.5, -1)
        a = eng.iteration(solver, 4000, 0, 2, iteration=0, seed=SEED, nchain=3).copy()
        cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]], obs=[0.0, [0.0, 0.0]], seed=SEED)
        meas = mci.Measure("if (idx < 0 || idx == 0) obs_add(0, rw[0]); if (idx < 0 || idx == 1) { obs_add(1, rw[1]); obs_add(2, rw[1] * 2.0); }")
        eng = mci.Engine(cfg, mci.catalog.sphere2(), measure=meas)
        eng.set_chain_speculation(lanes, 0.5, -1)
        b = eng.iteration(solver, 4000, 0, 2, iteration=0, seed=SEED, nchain=3).copy()
        res.append((a, b))
    for a, b in res[1:]:
        np.testing.assert_allclose(a, res[0][0], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(b, res[0][1], rtol=1e-9, atol=1e-300)


def test_automatic_group_size_follows_the_chain_count():
    """automatic: the largest group that keeps the launch within one wave per SIMD (65536 lanes), at least 8 lanes; more chains: one lane per chain"""
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]], seed=SEED)
    eng = mci.Engine(cfg, mci.catalog.x2y2())
    for nchain, want in ((1, 64), (64, 64), (128, 32), (512, 8), (1024, 1), (4096, 1)):   # (groups of 2 and 4 lanes lose to one lane per chain)
        eng.iteration("vegasmc", 8192, 0, 16, iteration=0, seed=SEED, nchain=nchain)
        assert eng.last_chain_speculation()[0] == want, (nchain, eng.last_chain_speculation())
    eng.set_chain_speculation(1)
    eng.iteration("vegasmc", 8192, 0, 16, iteration=0, seed=SEED, nchain=1)
    assert eng.last_chain_speculation() == (1, 0)


def test_the_campaign_case_the_compiler_got_wrong(oracle, monkeypatch):
    """Case 205 of the carried-chain campaign with a random group size and tree (tools/fuzz_layouts.py --carry --lanes): a composite pool of
    three leaves next to a Discrete pool nobody uses, ten draws.  ROCm 7.2's compiler turned the :vegasmc group kernel of this layout into
    one with the right chains and statistics and its histogram adds in the wrong bins; -opt-bisect-limit pins it on the backend's
    si-optimize-exec-masking-pre-ra, which the several-lanes-per-chain units are compiled without (csrc/mci_jit.h,
    profiles/r05_fuzz.txt)."""
    from layout_cases import check_carried_iterations
    monkeypatch.setenv("FUZZ_LANES", "1")
    check_carried_iterations(oracle, 205)


def test_a_new_group_code_object_proves_itself_before_it_is_trusted(oracle, tmp_path, monkeypatch):
    """mci_chain_speculation_status: the first launch through a several-lanes-per-chain code object that has never run on a device is
    preceded by a 2-block, 512-step run through it and through the lane-per-chain kernel (both step the reference's chain,
    vegas_mc/montecarlo.jl:198-211); agreement leaves a marker next to the code object and is never checked again.  The launch that
    triggered the check is the launch it would have been: same packed buffer as the oracle's chain, chain carry and logs untouched."""
    monkeypatch.setenv("MCI_KERNEL_CACHE", str(tmp_path))      # a cold cache: nothing in it has a marker
    for solver, osolver in (("vegasmc", oracle.VEGASMC), ("mcmc", oracle.MCMC)):
        cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]])
        eng = mci.Engine(cfg, mci.catalog.sphere2())
        assert eng.chain_speculation_status(solver) == 0
        got = eng.iteration(solver, 625, 0, 16, iteration=0, seed=SEED, nchain=1)
        assert eng.chain_speculation_status(solver) == 1 and eng.last_chain_speculation()[0] == 64 and eng.last_chain_launch() == (1, False)
        marker = eng.code_object(solver + "_lanes") + ".ok"
        assert os.path.exists(marker) and "hiprtc" in open(marker).read()
        ocfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[2], [3]])
        ref = ocfg.iteration(osolver, "sphere2", None, 625, 0, 16, 0, SEED, nchain=1)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
        eng.close()
        t = os.path.getmtime(marker)
        eng = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]]), mci.catalog.sphere2())
        eng.iteration(solver, 625, 0, 16, iteration=0, seed=SEED, nchain=1)
        assert eng.chain_speculation_status(solver) == 1 and os.path.getmtime(marker) == t      # (from the marker: not run again)
        eng.close()


def test_the_self_check_catches_the_miscompiled_group_kernel(tmp_path):
    """Case 205 with the backend pass that miscompiles it RE-ENABLED (MCI_JIT_FLAGS names the switch, so csrc/mci_jit.h leaves it alone):
    right chains, histogram adds in the wrong bins (profiles/r05_fuzz.txt).  The self-check of the new code object sees it: status -1,
    one warning, and the problem runs -- correctly -- with one lane per chain.  Without the check the launch returns the wrong histogram
    silently (second run: the check switched off).  Each run is a process of its own (tools/selfcheck_case205.py) on the ROCm
    installation's compiler: the miscompile is that compiler's -- the comgr PyTorch bundles, which a process that imported torch first
    compiles with, gets the layout right either way (profiles/r06_ablation.txt E)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCI_KERNEL_CACHE=str(tmp_path), MCI_JIT_FLAGS="-mllvm -amdgpu-opt-exec-mask-pre-ra=1", AMD_COMGR_CACHE="0")

    def run(*args):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "selfcheck_case205.py"), *args], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]), r.stderr
    out, err = run()
    assert "libamd_comgr.so.3" in out["compiler"], out["compiler"]         # (the installation's, not a bundled copy)
    if out["status"] == 1:
        pytest.skip("this toolchain compiles case 205 correctly with the pass on")
    assert out["status"] == -1 and out["lanes"] == 1 and "does not reproduce its lane-per-chain kernel" in err, (out, err[-2000:])
    assert out["stats_ok"] and out["hist_ok"]                       # the histogram too: the lane-per-chain kernel ran
    out, err = run("--no-check")                                    # what the check stands in front of
    assert out["status"] == 0 and out["lanes"] == 64 and "does not reproduce" not in err
    assert out["stats_ok"] and not out["hist_ok"] and out["hist_mismatches"] > 100      # right chains and statistics, the histogram in the wrong bins


def test_a_group_unit_that_does_not_compile_leaves_the_solver_on_one_lane_per_chain(oracle, tmp_path, monkeypatch, capfd):
    """The several-lanes-per-chain kernel is its own translation unit (up to 512 VGPRs, a backend switch a later compiler may refuse).
    Under automatic lanes a unit that fails to compile must not take the solver down: the lane-per-chain kernel steps the same chains
    (status -2, one note on stderr, the oracle's numbers).  Asked for explicitly (mci_set_chain_speculation(64)) the failure is the
    caller's to see: MCI_ERR_COMPILE.  (-DSpecLane=1 breaks mci_spec.h and nothing else.)"""
    monkeypatch.setenv("MCI_KERNEL_CACHE", str(tmp_path))
    monkeypatch.setenv("MCI_JIT_FLAGS", "-DSpecLane=1")
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]])
    eng = mci.Engine(cfg, mci.catalog.sphere2())
    got = eng.iteration("vegasmc", 625, 0, 16, iteration=0, seed=SEED, nchain=1)
    assert eng.chain_speculation_status("vegasmc") == -2 and eng.last_chain_speculation() == (1, 0)
    assert "did not compile; one lane per chain instead" in capfd.readouterr().err
    ocfg = oracle.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[2], [3]])
    ref = ocfg.iteration(oracle.VEGASMC, "sphere2", None, 625, 0, 16, 0, SEED, nchain=1)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-300)
    got = eng.iteration("vegasmc", 625, 0, 16, iteration=1, seed=SEED, nchain=1)       # ... and stays there without trying again
    assert eng.last_chain_speculation() == (1, 0) and "did not compile" not in capfd.readouterr().err
    eng.close()
    eng = mci.Engine(mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2], [3]]), mci.catalog.sphere2())
    eng.set_chain_speculation(64)
    with pytest.raises(mci.MCIError) as e:
        eng.iteration("vegasmc", 625, 0, 16, iteration=0, seed=SEED, nchain=1)
    assert "failed to compile" in str(e.value)
    eng.close()
