#!/usr/bin/env python3
"""Development tool (GPU box): is the slow start of a run the grid or the GPU?  40 launches of the C2 sample kernel on the SAME
(uniform, never adapted) grid, then 40 with adaptation, then 40 more without: per-launch HIP-event durations."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcintegration_jl_amd as mci
L = math.sqrt(50.0)
cfg = mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]], seed=1)
eng = mci.Engine(cfg, mci.catalog.gaussian(16))
eng.compile("vegas")
for label, adapt, n in (("uniform grid, adapt=False", False, 40), ("adapting", True, 40), ("adapted grid, adapt=False", False, 40), ("after 2 s idle, adapt=False", False, 20)):
    if label.startswith("after"):
        time.sleep(2.0)
    eng.integrate("vegas", neval=10**8, niter=n, block=16, seed=1, adapt=adapt, first_iteration=eng_it if (eng_it := getattr(eng, "_it", 0)) else 0)
    eng._it = getattr(eng, "_it", 0) + n
    ms, wg, th = eng.kernel_times_ms(n)
    print("%-28s" % label, " ".join("%.0f" % (1e3 * m) for m in ms))
