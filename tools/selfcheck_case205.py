#!/usr/bin/env python3
"""Development tool / test helper (GPU box): campaign case 205 -- the layout whose :vegasmc group kernel ROCm 7.2's
si-optimize-exec-masking-pre-ra miscompiles (profiles/r05_fuzz.txt) -- through the several-lanes-per-chain kernel in a process of its own
that pins the ROCm installation's compiler first (mci.use_rocm_compiler): a pytest process in which some test module has imported
PyTorch compiles with the comgr PyTorch bundles, and THAT compiler gets case 205 right with or without the pass.  With
MCI_JIT_FLAGS="-mllvm -amdgpu-opt-exec-mask-pre-ra=1" the pass is back on.  Prints one JSON line: the self-check's status, the lanes the
launch used, and whether statistics and histogram equal the oracle's.   usage: python tools/selfcheck_case205.py [--no-check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import mcintegration_jl_amd as mci
mci.use_rocm_compiler()
import mci_oracle as oracle
from layout_cases import random_case
from mcintegration_jl_amd._lib import check, lib

SEED = 20240229
if "--no-check" in sys.argv:
    check(lib().mci_debug_override(b"spec_self_check", 0, 1))
rng = np.random.default_rng(11000 + 205)
var, oleaves, dof, body, ndraw = random_case(rng)
oracle.build()
oracle.set_rng_rounds(10)
fn = oracle.compile_c_integrand(body)
ref = oracle.Config(oleaves, dof).iteration(oracle.VEGASMC, fn, None, 1200, 0, 2, 0, SEED, nchain=2)
nstat = 2 * len(dof) + 2 + len(dof) + 1
eng = mci.Engine(mci.Configuration(var=var, dof=dof, seed=SEED), mci.Integrand(body))
eng.set_chain_speculation(64, 0.5, 3)
got = eng.iteration("vegasmc", 1200, 0, 2, iteration=0, seed=SEED, nchain=2)
out = dict(compiler=mci.compiler_id(), status=eng.chain_speculation_status("vegasmc"), lanes=eng.last_chain_speculation()[0],
           stats_ok=bool(np.allclose(got[:nstat], ref[:nstat], rtol=1e-8, atol=1e-300)),
           hist_ok=bool(np.allclose(got[nstat:], ref[nstat:], rtol=1e-7)),
           hist_mismatches=int((~np.isclose(got[nstat:], ref[nstat:], rtol=1e-7)).sum()), hist_entries=int(got.size - nstat))
eng.close()
mci.shutdown()
print(json.dumps(out))
