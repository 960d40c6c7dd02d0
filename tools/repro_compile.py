#!/usr/bin/env python3
"""Development tool (no GPU): compile the several-lanes-per-chain :vegasmc unit of case 205 of the carried-chain campaign into the kernel
cache, under the options in MCI_JIT_FLAGS -- e.g. MCI_JIT_FLAGS="-mllvm -opt-bisect-limit=55582" -- so that tools/bisect_passes.sh can run
the variants on a GPU box (the cache travels with the repository).  Prints the code object's path."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import mcintegration_jl_amd as mci
from layout_cases import random_case
rng = np.random.default_rng(11000 + 205)
var, oleaves, dof, body, ndraw = random_case(rng)
cfg = mci.Configuration(var=var, dof=dof, seed=1)
eng = mci.Engine(cfg, mci.Integrand(body), device=-1)
for unit in sys.argv[1:] or ["vegasmc_lanes"]:
    eng.compile(unit)
    print(eng.code_object(unit))
eng.close()
