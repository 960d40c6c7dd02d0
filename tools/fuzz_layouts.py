#!/usr/bin/env python3
"""Development tool (GPU box): randomised-layout parity campaign, wider than tests/test_hip_random_configs.py.

Every case draws a layout (1-5 pools of Continuous / Discrete / CompositeVar, 1-4 integrands with ragged dof tables, grids of 17 to
2000 increments), a launch shape (blocks, steps per block, chains per block, measure cadence, iteration number) and a generator
(Philox4x32-10 or -7), JIT-compiles the three sample-batch kernels for it and compares one iteration of each solver with the oracle on
the same Philox streams: packed sums and histograms to 1e-9 relative, holding-time histogram bucket by bucket, then a three-iteration
:vegas run with train! in between.  Failures are collected, not fatal.   usage: fuzz_layouts.py [--pipe | --persist | --carry | --walk] [--lanes] [first_case] [ncases]
(--lanes, with the default and the --carry campaign: a random number of lanes per chain and a random speculation tree per case, csrc/mci_spec.h)
(--pipe: layouts of the pipelined :vegas loop only; --persist: whole integrate() calls over one Continuous variable type run as ONE
persistent launch, against the oracle's loop; --carry: four consecutive iterations of :vegasmc and :mcmc with carried chains -- :mcmc:
resampled to the moved reweight factors -- and a changing chain count, doReweight! and train! in between; --walk: deterministic runs under train!'s
serial walk as slots with given decisions and under the general form of the recurrence, bit for bit)"""
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


from layout_cases import check_carried_iterations, check_persistent_call, check_walks_agree, pipe_case, random_case, vary_chain_lanes  # noqa: E402  (shared with tests/test_hip_steady_state.py, test_hip_persistent.py)


PIPE_MODE = False
PERSIST_MODE = False
CARRY_MODE = False
WALK_MODE = False
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
    what += vary_chain_lanes(eng, case_id)
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
    if "--carry" in sys.argv:     # four consecutive iterations of both chain solvers with carried chains and changing chain counts
        sys.argv.remove("--carry")
        CARRY_MODE = True
    if "--lanes" in sys.argv:     # a random group size and speculation tree per case (the default and the --carry campaign)
        sys.argv.remove("--lanes")
        os.environ["FUZZ_LANES"] = "1"
    if "--walk" in sys.argv:      # serial walk with given decisions against its general form, deterministic runs, bit for bit
        sys.argv.remove("--walk")
        WALK_MODE = True
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    oracle.build()
    bad, t0 = [], time.time()
    nslots = ngeneral = 0
    for c in range(first, first + n):
        try:
            w = run_persist_case(c) if PERSIST_MODE else check_carried_iterations(oracle, c, SEED) if CARRY_MODE else check_walks_agree(c, SEED) if WALK_MODE else run_case(c)
            print("ok   " + w, flush=True)
            if WALK_MODE:
                nslots += int(w.split("walks as slots ")[1].split(",")[0])
                ngeneral += int(w.split("general ")[1])
        except Exception as e:  # collect and go on
            bad.append(c)
            print("FAIL case %d: %s" % (c, "".join(traceback.format_exception_only(type(e), e))[:3000]), flush=True)
    oracle.set_rng_rounds(10)
    print("%d cases, %d failed %s in %.0f s" % (n, len(bad), bad, time.time() - t0) + (" (%d ran as one persistent launch)" % npersist if PERSIST_MODE else "")
          + (" (serial walks: %d as slots with given decisions, %d through the general form)" % (nslots, ngeneral) if WALK_MODE else ""))
    mci.shutdown()
