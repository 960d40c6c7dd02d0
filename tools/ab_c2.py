#!/usr/bin/env python3
"""Development tool (GPU box): A/B of compile-time variants (MCI_JIT_FLAGS) of the :vegas sample kernel on a BASELINE
workload: median HIP-event kernel time over 8 launches of 1e8 samples, after 24 launches of warm-up.  usage: ab_c2.py [c2|c2i|c4|c5v|g8|g12] 'flags' 'flags' ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, math, json
sys.path.insert(0, %r)
import numpy as np
import mcintegration_jl_amd as mci
from mcintegration_jl_amd import isa_mix
L = math.sqrt(50.0)
which = sys.argv[1]
if which == "c2":
    cfg, f = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1), mci.catalog.gaussian(16)
elif which == "c2i":
    cfg, f = mci.Configuration(var=mci.Continuous([(-L, L)] * 16), dof=[[1]], seed=1), mci.catalog.gaussian(16)
elif which in ("g8", "g12"):     # the headline integrand in 8 / 12 dimensions (lighter kernels: where the histogram-copy rule flips)
    d = int(which[1:])
    cfg, f = mci.Configuration(var=mci.Continuous(-L, L), dof=[[d]], seed=1), mci.catalog.gaussian(d)
elif which == "c4":
    cfg, f = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * 32), dof=[[1]], seed=1), mci.catalog.genz_product_peak(32)
else:
    cfg, f = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]], seed=1), mci.catalog.nested_gauss()
import os
eng = mci.Engine(cfg, f, rng_bits=int(os.environ.get("MCI_AB_RNG_BITS", "52")), rng_rounds=int(os.environ.get("MCI_AB_RNG_ROUNDS", "10")))
eng.integrate("vegas", neval=10**8, niter=24, block=16, seed=1)          # train + let the GPU come out of idle (~16 launches, tools/ramp_probe.py)
r = eng.integrate("vegas", neval=10**8, niter=8, block=16, seed=2, first_iteration=24, ignore=0)
ms, wg, th = eng.kernel_times_ms(8)
res = isa_mix.resources(eng.code_object("vegas"))["mci_vegas_batch"]
print(json.dumps(dict(kernel_ms=float(np.median(ms)), iter_ms=r["seconds"] / 8 * 1e3, wg=wg, threads=th, vgpr=res["vgpr"], spill=res["vgpr_spill"],
                      mean=float(r["mean"][0]), sigma=float(r["stdev"][0]))))
''' % ROOT

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    variants = sys.argv[2:] or [""]
    for i, flags in enumerate(variants):
        env = dict(os.environ)
        env["MCI_JIT_FLAGS"] = flags
        env["MCI_KERNEL_CACHE"] = "/tmp/mci_ab_cache"
        out = subprocess.run([sys.executable, "-c", CHILD, which], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else "ERROR " + out.stderr[-300:]
        print("%-4s %-52s %s" % (which, flags or "(default)", line), flush=True)
