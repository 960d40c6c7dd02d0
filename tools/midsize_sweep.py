#!/usr/bin/env python3
"""Development tool (GPU box): launch geometry of mid-size :vegas iterations (3e5 .. 3e6 samples, the sizes the reference's own tests
and examples run, test/montecarlo.jl:298-387): us per iteration of the library loop for the automatic geometry and for forced
(threads, workgroups per block) pairs.   usage: python tools/midsize_sweep.py [case ...]   cases: x2y2 gauss6 gauss16"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci

L = math.sqrt(50.0)
CASES = {
    "x2y2": (lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]], seed=1), mci.catalog.x2y2),
    "gauss6": (lambda: mci.Configuration(var=mci.Continuous(-L, L), dof=[[6]], seed=1), lambda: mci.catalog.gaussian(6)),
    "gauss16": (lambda: mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1), lambda: mci.catalog.gaussian(16)),
}
GEOM = [(None, None), (256, 8), (256, 16), (512, 4), (512, 8), (512, 16), (1024, 4), (1024, 8)]
if os.environ.get("SWEEP_GEOM"):   # e.g. SWEEP_GEOM="1024x16,512x32": other (threads x workgroups per block) pairs, block = 16
    GEOM = [(None, None)] + [tuple(int(v) for v in g.split("x")) for g in os.environ["SWEEP_GEOM"].split(",")]


def one(name, neval, threads, wpb, niter=60):
    mk, f = CASES[name]
    kw = {}
    if threads:
        kw = dict(threads=threads, wg_per_block=wpb)
    eng = mci.Engine(mk(), f(), **kw)
    eng.set_persistent("off")
    eng.integrate("vegas", neval=neval, niter=6, block=16, seed=1)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        r = eng.integrate("vegas", neval=neval, niter=niter, block=16, seed=1, first_iteration=6 + rep * niter, ignore=0)
        best = min(best, (time.perf_counter() - t0) / niter * 1e6)
    wg, th = eng.kernel_times_ms(1)[1:]
    eng.close()
    return best, wg, th, r["mean"][0], r["stdev"][0]


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for name in names:
        for neval in (3 * 10**5, 10**6, 3 * 10**6):
            row = []
            for threads, wpb in GEOM:
                us, wg, th, m, e = one(name, neval, threads, wpb)
                row.append("%s: %.1f us (wg=%d th=%d)" % ("auto" if threads is None else "T=%d wpb=%d" % (threads, wpb), us, wg, th))
            print("%-8s neval=%-8d %s" % (name, neval, " | ".join(row)), flush=True)
    mci.shutdown()
