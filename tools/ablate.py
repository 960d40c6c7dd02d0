#!/usr/bin/env python3
"""Development tool (runs on the GPU box): kernel-time attribution of mci_vegas_batch on the C2 workload
by compile-time ablation (MCI_JIT_FLAGS) and launch-geometry sweep.  Not part of the product path."""
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
L = math.sqrt(50.0)
threads, wpb, neval = int(sys.argv[1]), int(sys.argv[2]), int(float(sys.argv[3]))
cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1)
eng = mci.Engine(cfg, mci.catalog.gaussian(16), threads=threads, wg_per_block=wpb)
eng.integrate("vegas", neval=neval, niter=4, block=16, seed=1)           # warm-up + train
eng.integrate("vegas", neval=neval, niter=5, block=16, seed=2, first_iteration=4)
ms, wg, th = eng.kernel_times_ms(5)
print(json.dumps(dict(ms=float(np.median(ms)), wg=wg, threads=th)))
''' % ROOT


def run(flags, threads=256, wpb=0, neval=1e8):
    env = dict(os.environ)
    env["MCI_JIT_FLAGS"] = flags
    env["MCI_KERNEL_CACHE"] = "/tmp/mci_ablate_cache"
    out = subprocess.run([sys.executable, "-c", CHILD, str(threads), str(wpb), str(neval)], env=env, capture_output=True, text=True)
    if out.returncode:
        return dict(error=out.stderr[-400:])
    return json.loads(out.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    rows = []
    for name, flags in [("baseline", ""), ("cheap rng", "-DMCI_ABL_CHEAPRNG"), ("no hist", "-DMCI_ABL_NOHIST"),
                        ("no table", "-DMCI_ABL_NOTABLE"), ("no hist+table", "-DMCI_ABL_NOHIST -DMCI_ABL_NOTABLE"),
                        ("cheap rng + no hist + no table", "-DMCI_ABL_CHEAPRNG -DMCI_ABL_NOHIST -DMCI_ABL_NOTABLE"),
                        ("philox7", "-DMCI_PHILOX_ROUNDS=7"), ("fp-contract fast", "-ffp-contract=fast")]:
        r = run(flags)
        rows.append((name, r))
        print("%-34s %s" % (name, r), flush=True)
    for threads, wpb in [(64, 0), (128, 0), (512, 0), (1024, 0), (256, 32), (256, 128), (256, 256), (512, 64), (1024, 16), (1024, 32)]:
        r = run("", threads, wpb)
        print("threads=%-5d wpb=%-4d %s" % (threads, wpb, r), flush=True)
