#!/usr/bin/env python3
"""Development tool (GPU box): one case of the carried-chain campaign (tests/layout_cases.py check_carried_iterations), first :vegasmc iteration,
under several lane settings: where do packed sums / histograms differ from the oracle?   usage: python tools/repro_case.py <case_id>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import mcintegration_jl_amd as mci
import mci_oracle as oracle
from layout_cases import random_case

case_id = int(sys.argv[1])
seed = 20260930
rng = np.random.default_rng(11000 + case_id)
var, oleaves, dof, body, ndraw = random_case(rng)
nblk = int(rng.integers(1, 5))
npb = int(rng.choice([1200, 2400, 4800]))
counts = [int(rng.choice([2, 5, 8, 16, 40])) for _ in range(4)]
mfreq = int(rng.choice([1, 1, 3]))
if os.environ.get("R_NCHAIN"): counts[0] = int(os.environ["R_NCHAIN"])
if os.environ.get("R_MFREQ"): mfreq = int(os.environ["R_MFREQ"])
if os.environ.get("R_NPB"): npb = int(os.environ["R_NPB"])
if os.environ.get("R_NBLK"): nblk = int(os.environ["R_NBLK"])
if os.environ.get("R_DOF"): dof = eval(os.environ["R_DOF"])
print("body:", body)
print("dof", dof, "ndraw", ndraw, "nblk", nblk, "npb", npb, "counts", counts, "mfreq", mfreq)
print("leaves", oleaves)
oracle.set_rng_rounds(10)
fn = oracle.compile_c_integrand(body)
ni = len(dof)
ocfg = oracle.Config(oleaves, dof)
ref = ocfg.iteration(oracle.VEGASMC, fn, None, npb, 0, nblk, 0, seed, measurefreq=mfreq, nchain=counts[0])
TREES = ((1, 0.0, -1), (8, 1e-3, -1), (64, 0.5, 3))
if os.environ.get("R_ONLY_LANES"): TREES = tuple(t for t in TREES if t[0] == int(os.environ["R_ONLY_LANES"]))
for lanes, accept, limit in TREES:
    cfg = mci.Configuration(var=var, dof=dof, seed=seed)
    eng = mci.Engine(cfg, mci.Integrand(body))
    eng.set_chain_speculation(lanes, accept, limit)
    n = eng.nobs
    nstat = 2 * n + 2 + ni + 1
    got = eng.iteration("vegasmc", npb, 0, nblk, iteration=0, seed=seed, measurefreq=mfreq, nchain=counts[0])
    hs, hr = got[nstat:], ref[nstat:]
    print("   stats got", got[:nstat], "\n   stats ref", ref[:nstat])
    bad = np.nonzero(~np.isclose(hs, hr, rtol=1e-7, atol=0))[0]
    print("lanes", lanes, accept, limit, "-> used", eng.last_chain_speculation(), "launch (wg, threads)", eng.kernel_times_ms(1)[1:], " stats max rel", np.max(np.abs(got[:nstat] - ref[:nstat]) / np.maximum(np.abs(ref[:nstat]), 1e-300)),
          " hist mismatches", bad.size, "of", hs.size, " nonfinite got", int((~np.isfinite(hs)).sum()), "ref", int((~np.isfinite(hr)).sum()))
    if bad.size:
        i = bad[:6]
        print("    idx", i, "got", hs[i], "ref", hr[i])
        print("    got min/max", hs.min(), hs.max(), " ref min/max", hr.min(), hr.max())
        off = 0
        for li, lf in enumerate(oleaves):   # which leaves' histograms differ
            nb = (lf.get("npts") or 1000) - 1 if lf["kind"] == 0 else int(lf["upper"] - lf["lower"] + 1)
            seg_g, seg_r = hs[off:off + nb], hr[off:off + nb]
            nbad = int((~np.isclose(seg_g, seg_r, rtol=1e-7, atol=0)).sum())
            print("      leaf %d kind %d pool %d bins %d: mismatches %d  sum got %.6g ref %.6g  argmax got %d ref %d" % (li, lf["kind"], lf["pool"], nb, nbad, seg_g.sum(), seg_r.sum(), int(seg_g.argmax()), int(seg_r.argmax())))
            off += nb
        print("      total bins", off, "of", hs.size)
    eng.close()
mci.shutdown()
