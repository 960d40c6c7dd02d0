"""Bias of the chain solvers on the bubble (C3) against the finite-temperature polarisation, averaged over seeds.
usage: [MCI_CHAIN_RESAMPLE=K] python tools/chain_start_bias.py '[["mcmc", 1e8, 0, 16, 6], ...]'   (solver, neval, nchain, block, nseeds)"""
import json, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mcintegration_jl_amd as mci
from catalog_params import bubble_exact_finite_T

ft = np.array(bubble_exact_finite_T())
p = mci.catalog.bubble_parameters()
for solver, ne, nchain, block, nseeds in json.loads(sys.argv[1]):
    allm, alle, secs = [], [], 0.0
    for seed in range(1, int(nseeds) + 1):
        var = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, math.pi, alpha=3.0), mci.Continuous(0.0, 2 * math.pi, alpha=3.0),
               mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
        cfg = mci.Configuration(var=var, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)])
        eng = mci.Engine(cfg, mci.catalog.bubble(), measure=mci.bin_by(4))
        eng.compile(solver)
        eng.integrate(solver, neval=int(ne), niter=5, block=int(block), seed=seed, nchain=int(nchain))
        r = eng.integrate(solver, neval=int(ne), niter=10, block=int(block), seed=seed, first_iteration=5, ignore=0, nchain=int(nchain))
        allm.append(r["mean"]); alle.append(r["stdev"]); secs += r["seconds"]
    allm, alle = np.array(allm), np.array(alle)
    err = np.sqrt((alle ** 2).sum(0)) / len(allm)
    print("K=%-5s %-8s neval=%.0e nchain=%-6d block=%-3d seeds=%d  %.2f s/run  bias=%s  bias/err=%s  (scatter-based err %s)" % (
        os.environ.get("MCI_CHAIN_RESAMPLE", "0"), solver, ne, nchain, block, nseeds, secs / len(allm),
        np.array2string(allm.mean(0) - ft, precision=2), np.round((allm.mean(0) - ft) / err, 2),
        np.array2string(allm.std(0, ddof=1) / math.sqrt(len(allm)), precision=2)), flush=True)
