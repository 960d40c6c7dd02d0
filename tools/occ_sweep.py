#!/usr/bin/env python3
"""Development tool (GPU box): forced-occupancy / launch-geometry sweep of the headline kernel (C2 shared pool)."""
import json, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, math
sys.path.insert(0, %r)
import numpy as np
import mcintegration_jl_amd as mci
threads, wpb = int(sys.argv[1]), int(sys.argv[2])
L = math.sqrt(50.0)
cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1)
eng = mci.Engine(cfg, mci.catalog.gaussian(16), threads=threads, wg_per_block=wpb)
eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=1)
eng.integrate("vegas", neval=10**8, niter=5, block=16, seed=2, first_iteration=5)
ms, wg, th = eng.kernel_times_ms(5)
print(json.dumps(dict(ms=float(np.median(ms)), Gs=1e8 / float(np.median(ms)) / 1e6, wg=wg, threads=th)))
''' % ROOT

def run(threads, wpb, flags=""):
    env = dict(os.environ)
    env["MCI_KERNEL_CACHE"] = "/tmp/mci_occ_cache"
    if flags: env["MCI_JIT_FLAGS"] = flags
    out = subprocess.run([sys.executable, "-c", CHILD, str(threads), str(wpb)], env=env, capture_output=True, text=True)
    if out.returncode: return dict(error=out.stderr[-300:])
    return json.loads(out.stdout.strip().splitlines()[-1])

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "occ"
    if which == "occ":
        for flags in ("", "-DMCI_WAVES=4", "-DMCI_WAVES=5", "-DMCI_WAVES=6", "-DMCI_WAVES=8"):
            print("threads=256 wpb=auto flags=%-16s %s" % (flags, run(256, 0, flags)), flush=True)
        for threads, wpb in ((256, 64), (256, 256), (512, 64), (128, 256), (1024, 32)):
            print("threads=%d wpb=%d %s" % (threads, wpb, run(threads, wpb)), flush=True)
    else:
        for flags in ("", "-DMCI_DRAW_FENCE=1", "-DMCI_DRAW_FENCE=2", "-DMCI_DRAW_FENCE=4", "-mllvm -amdgpu-schedule-metric-bias=0", "-mllvm -amdgpu-use-amdgpu-trackers=1"):
            print("threads=256 flags=%-44s %s" % (flags, run(256, 0, flags)), flush=True)
