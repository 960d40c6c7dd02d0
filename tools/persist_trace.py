#!/usr/bin/env python3
"""Development tool (GPU box): where a turn of the persistent :vegas launch spends its time.  Builds the kernel with
MCI_PERSIST_TRACE (wall-clock stamps, 10 ns, of workgroups 0 / nleaf / G-1 at every phase) and prints the phase durations in us."""
import ctypes as C
import os
import sys

os.environ["MCI_JIT_FLAGS"] = (os.environ.get("MCI_JIT_FLAGS", "") + " -DMCI_PERSIST_TRACE=1").strip()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcintegration_jl_amd as mci
from mcintegration_jl_amd import _lib

if __name__ == "__main__":
    neval = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000
    cfg = mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[2]])
    eng = mci.Engine(cfg, mci.catalog.x2y2())
    if len(sys.argv) > 2:
        eng.set_launch(int(sys.argv[2]), 0)
    eng.set_persistent("on")
    eng.integrate("vegas", neval=neval, niter=3, block=16, seed=1)
    eng.integrate("vegas", neval=neval, niter=8, block=16, seed=1, first_iteration=3)
    n = 8 + 3 * 8 * 8 + 16
    buf = (C.c_uint64 * n)()
    L = _lib.lib()
    assert L.mci_debug_persist_words(eng.p, buf, n) == 0
    raw = np.array(list(buf), dtype=np.uint64).astype(np.int64)
    w = raw[8:8 + 192].reshape(3, 8, 8)
    tt = raw[8 + 192:8 + 192 + 8]
    sc = raw[8 + 192 + 8:8 + 192 + 10]
    print('shader clock over turns 0..7: %.0f MHz' % ((int(sc[1]) - int(sc[0])) / ((int(w[0, 7, 0]) - int(w[0, 0, 0])) / 100.0)))
    t00 = w[:, 0, 0].min()
    names = ["start", "sampled+flushed", "signalled", "all arrived", "refined / merged", "signalled done", "-"]
    for g, who in enumerate(["workgroup 0 (samples, refines, writes the map back)", "statistics workgroup", "workgroup G-1 (samples, refines)"]):
        print(who)
        for it in range(8):
            row = w[g, it]
            ts = [(int(row[k]) - int(t00)) / 100.0 if row[k] else None for k in range(7)]
            print("  turn %d: " % it + "  ".join("%s %s" % (names[k], "%.2f" % ts[k] if ts[k] is not None else "-") for k in range(7)))
    tn = ["enter", "histogram checked", "smoothed", "sum16", "rescaled", "prefix scan", "bisection + new grid", "cleared"]
    print("train_leaf of workgroup 0, turn 7 (us from entry): " + "  ".join("%s %.2f" % (tn[k], (int(tt[k]) - int(tt[0])) / 100.0) for k in range(8)))
