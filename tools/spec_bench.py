#!/usr/bin/env python3
"""Development tool (GPU box): several lanes per chain (csrc/mci_spec.h) against one lane per chain.

    python tools/spec_bench.py steps <case> <solver> [neval] [nchain] [block]
        kernel time of ONE iteration of a chain solver on a trained map, for lane-per-chain and for groups of 64 lanes along trees built for
        several acceptances / accept-edge limits; the measured acceptance of the chain (config.propose / config.accept) next to it
    python tools/spec_bench.py default [niter]
        the reference's default call -- integrate(solver = :vegasmc | :mcmc, neval = 1e4, niter = 10, block = 16) -- per iteration
cases: tools/mcmc_policy.py (c5 | bubble | cos | x2 | sphere2 | hyper | log)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mcintegration_jl_amd as mci
from mcmc_policy import case


def acceptance(eng, solver):
    pr, ac = eng.acceptance()
    tot_p, tot_a = pr.sum(), ac.sum()
    return tot_a / max(tot_p, 1e-300), tot_p


def steps(name, solver, neval, nchain, block):
    trees = [(1, 0.0, -1)] + [(64, a, m) for a, m in ((1e-3, -1), (0.2, 1), (0.35, 1), (0.35, 2), (0.5, 2), (0.5, 3), (0.5, -1), (0.7, -1))] + [(16, 0.5, -1), (4, 0.5, -1)]
    if solver == "vegasmc":
        trees = [t for t in trees if t[2] in (-1, 1, 2)]
    print("%s :%s neval=%.0e block=%d nchain=%d per block (trained map: 5 iterations first); one iteration, HIP-event kernel time" % (name, solver, neval, block, nchain))
    for lanes, accept, limit in trees:
        cfg, f, meas, exact = case(name)
        eng = mci.Engine(cfg, f, measure=meas)
        eng.set_chain_carry("off")
        eng.set_chain_speculation(1)
        eng.integrate(solver, neval=neval, niter=5, block=block, seed=1, nchain=max(nchain, 64))   # train the map with plenty of chains
        eng.set_chain_speculation(lanes, accept, limit)
        eng.set_kernel_timing(1)
        npb = neval // block
        best, it = 1e30, 5
        for rep in range(3):
            eng.run(solver, npb, 0, block, it, 1, 1, nchain, 0.1)
            ms = float(eng.kernel_times_ms(1)[0][-1])
            acc, prop = acceptance(eng, solver)
            eng.finish(solver, block, False, 1.0)
            best = min(best, ms)
            it += 1
        g, mx = eng.last_chain_speculation()
        print("  lanes %2d  tree(accept=%-5g limit=%2d -> %d accept levels)  kernel %9.3f ms  %8.3f us per chain step  (accepted / proposed %.3f, proposed / steps %.3f)" % (
            g, accept, limit, mx, best, best * 1e3 / (npb / nchain), acc, prop / (block * npb)), flush=True)
        eng.close()


def default_call(niter):
    for solver in ("vegasmc", "mcmc"):
        for lanes in (1, -1):
            cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
            eng = mci.Engine(cfg, mci.catalog.x2y2())
            eng.set_chain_speculation(lanes)
            eng.integrate(solver, neval=10**4, niter=3, block=16, seed=1)
            t0 = time.perf_counter()
            reps = 20
            for r in range(reps):
                res = eng.integrate(solver, neval=10**4, niter=niter, block=16, seed=1, first_iteration=3 + r * niter)
            dt = (time.perf_counter() - t0) / reps
            g = eng.last_chain_speculation()
            print("default call :%s lanes=%s -> %s  %.1f us per call, %.1f us per iteration; mean %.5f +- %.1e" % (
                solver, lanes, g, dt * 1e6, dt / niter * 1e6, res["mean"][0], res["stdev"][0]), flush=True)
            eng.close()


if __name__ == "__main__":
    if sys.argv[1] == "steps":
        steps(sys.argv[2], sys.argv[3], int(float(sys.argv[4])) if len(sys.argv) > 4 else 10**6, int(sys.argv[5]) if len(sys.argv) > 5 else 1,
              int(sys.argv[6]) if len(sys.argv) > 6 else 16)
    else:
        default_call(int(sys.argv[2]) if len(sys.argv) > 2 else 10)
    mci.shutdown()
