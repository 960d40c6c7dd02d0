#!/usr/bin/env python3
"""Development tool (GPU box): launch-geometry sweep for the 32-grid Genz config (C4) and the 16-grid Gaussian."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import mcintegration_jl_amd as mci
threads, wpb, neval, D = int(sys.argv[1]), int(sys.argv[2]), int(float(sys.argv[3])), int(sys.argv[4])
cfg = mci.Configuration(var=mci.Continuous([(0.0, 1.0)] * D), dof=[[1]], seed=1)
eng = mci.Engine(cfg, mci.catalog.genz_product_peak(D), threads=threads, wg_per_block=wpb)
eng.integrate("vegas", neval=neval, niter=3, block=16, seed=1)
eng.integrate("vegas", neval=neval, niter=3, block=16, seed=2, first_iteration=3)
ms, wg, th = eng.kernel_times_ms(3)
print(json.dumps(dict(ms=float(np.median(ms)), Gs=neval / float(np.median(ms)) / 1e6, wg=wg, threads=th, mode=eng.table_mode, lds=eng.lds_bytes)))
''' % ROOT

def run(threads, wpb, tile_bins=None, flags="", D=32, neval=1e8):
    env = dict(os.environ)
    env["MCI_KERNEL_CACHE"] = "/tmp/mci_c4_cache"
    if tile_bins: env["MCI_HIST_TILE_BINS"] = str(tile_bins)
    if flags: env["MCI_JIT_FLAGS"] = flags
    out = subprocess.run([sys.executable, "-c", CHILD, str(threads), str(wpb), str(neval), str(D)], env=env, capture_output=True, text=True)
    if out.returncode: return dict(error=out.stderr[-300:])
    return json.loads(out.stdout.strip().splitlines()[-1])

if __name__ == "__main__":
    for D in (32, 16):
        for flags in ("", "-DMCI_DRAW_FENCE=2", "-DMCI_DRAW_FENCE=4"):
            print("D=%d threads=512 flags=%-36s %s" % (D, flags, run(512, 0, None, flags, D)), flush=True)
