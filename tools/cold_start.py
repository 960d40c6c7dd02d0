#!/usr/bin/env python3
"""Cold-start latency of a NEW integrand body (nothing in the kernel cache): hiprtc time per solver, and -- with a GPU -- the wall time
of the first integrate() call.  The reference's counterpart is Julia's JIT specialising `montecarlo` on the closure
(example/benchmark/cuba/benchmark.jl:146-147: 0.246 s for the whole Cuba-11 run, 9 % of it compile time).
usage: python tools/cold_start.py [--offline]"""
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MCI_KERNEL_CACHE"] = tempfile.mkdtemp(prefix="mci_cold_")
import numpy as np
import mcintegration_jl_amd as mci

OFFLINE = "--offline" in sys.argv or mci.engine.device_count() == 0
L = math.sqrt(50.0)
PI = math.pi
p = mci.catalog.bubble_parameters()


def cases(tag):
    bub = (mci.Continuous(0.0, 1.0, alpha=3.0), mci.Continuous(0.0, PI, alpha=3.0), mci.Continuous(0.0, 2 * PI, alpha=3.0),
           mci.Continuous(0.0, p["beta"], alpha=3.0), mci.Discrete(1, 4, adapt=False))
    # (every body gets a unique constant so that nothing can come from a cache)
    return [
        ("1-D user body", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[1]]), mci.Integrand("w[0] = log(x[0]) / sqrt(x[0]) + %s;" % tag), None),
        ("C2 16-D Gaussian", lambda: mci.Configuration(var=mci.Continuous(-L, L), dof=[[16]]),
         mci.Integrand(mci.catalog.gaussian(16).body + "\nw[0] += %s;" % tag, [16.0]), None),
        ("C3 bubble", lambda: mci.Configuration(var=bub, dof=[[1, 1, 1, 1, 1]], obs=[np.zeros(4)]),
         mci.Integrand(mci.catalog.bubble().body + "\nw[0] += %s;" % tag, mci.catalog.bubble().userdata), mci.bin_by(4)),
        ("C5 nested Gaussians", lambda: mci.Configuration(var=mci.Continuous(0.0, 1.0), dof=[[3], [6], [9], [12]]),
         mci.Integrand(mci.catalog.nested_gauss().body + "\nw[0] += %s;" % tag, mci.catalog.nested_gauss().userdata), None),
    ]


if "--first-call" not in sys.argv:
    print("%-22s %-8s %10s" % ("integrand", "solver", "hiprtc s"))
for solver in ("vegas", "vegasmc", "mcmc") if "--first-call" not in sys.argv else ():
    for name, mk, f, meas in cases("1e-300 * %d" % (hash(solver) % 1000)):
        eng = mci.Engine(mk(), f, measure=meas, device=-1 if OFFLINE else 0)
        t0 = time.perf_counter()
        eng.compile(solver)
        print("%-22s %-8s %10.3f" % (name, solver, time.perf_counter() - t0), flush=True)
        eng.close()
if not OFFLINE and "--first-call" in sys.argv:   # (child process: one solver's first call, printed as one line)
    solver = sys.argv[sys.argv.index("--first-call") + 1]
    name, mk, f, meas = cases("2e-300 * %d" % (hash(solver) % 1000))[0]
    t0 = time.perf_counter()
    r = mci.integrate(f, config=mk(), solver=solver, neval=1e4, measure=meas, seed=1)
    t1 = time.perf_counter()
    r = mci.integrate(f, config=mk(), solver=solver, neval=1e4, measure=meas, seed=1)
    t2 = time.perf_counter()
    # (a second new body right after: nothing queues in front of its compile -- the persistent :vegas kernel's translation unit, 0.8 s of
    # hiprtc on a thread of its own, starts only after 256 launch-bound calls of a process; when it was started by the second call this
    # line read 0.69 s under :vegas, comgr serialising the two compiles)
    name2, mk2, f2, meas2 = cases("3e-300 * %d" % (hash(solver) % 1000))[0]
    t3 = time.perf_counter()
    mci.integrate(f2, config=mk2(), solver=solver, neval=1e4, measure=meas2, seed=1)
    t4 = time.perf_counter()
    print("%-22s %-8s first %.3f s   again (code object cached, new engine) %.4f s   a second new body right after %.3f s   mean %s" % (
        name, solver, t1 - t0, t2 - t1, t4 - t3, r.mean), flush=True)
elif not OFFLINE:
    print("\nfirst integrate() of a new body in a fresh process (engine creation + JIT + 10 iterations of neval = 1e4), the same call again,\n"
          "and a second new body right after:", flush=True)
    import subprocess
    for solver in ("vegas", "vegasmc", "mcmc"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--first-call", solver], check=False)
mci.shutdown()
