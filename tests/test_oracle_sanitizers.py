"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers).  Every parity claim leans on
oracle/mci_oracle.c, so its known-answer suite and one whole `mcio_integrate` per solver (carried chains, several chains per block, a
Discrete and a composite pool, complex weights) run once against an instrumented build: oracle/Makefile `sanitize`, loaded into a
Python started with LD_PRELOAD=libasan.so.  GPU AddressSanitizer is not available on this pool; this is the CPU side only."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORACLE = os.path.join(ROOT, "oracle")

DRIVER = r"""
import sys
sys.path.insert(0, %(oracle)r)
import numpy as np
import mci_oracle as O
assert O._SO.endswith("libmci_oracle_san.so"), O._SO
# one whole run per solver (grid -> histogram -> grid, doReweight!, chains carried over iterations, many chains per block)
cases = [(O.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0)], [[2], [3]]), "sphere2"),
         (O.Config([dict(kind=0, pool=0, lower=0.0, upper=1.0), dict(kind=1, pool=1, lower=1, upper=3)], [[2, 1], [3, 0]]),
          O.compile_c_integrand("w[0] = x[0] * x[1] * x[3]; w[1] = x[0] + x[1] * x[2];", ni=2))]
for cfg, f in cases:
    for solver, nchain in ((O.VEGAS, 1), (O.VEGASMC, 1), (O.VEGASMC, 4), (O.MCMC, 1), (O.MCMC, 4)):
        r = cfg.integrate(solver, f, None, 8000, niter=4, block=4, seed=7, nchain=nchain)
        assert r["rc"] == 0 and np.all(np.isfinite(r["mean"])) and np.all(np.isfinite(r["stdev"])), (solver, nchain, r)
print("sanitized runs ok")
"""


@pytest.fixture(scope="module")
def san_env():
    so = os.path.join(ORACLE, "libmci_oracle_san.so")
    subprocess.check_call(["make", "-C", ORACLE, "-s", "sanitize"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan.so")
    env = dict(os.environ, LD_PRELOAD=asan, MCI_ORACLE_SO=so, OMP_NUM_THREADS="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return env


def test_known_answer_suite_under_asan_and_ubsan(san_env):
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(HERE, "test_oracle_known_answers.py")],
                         env=san_env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]


def test_whole_runs_of_every_solver_under_asan_and_ubsan(san_env):
    out = subprocess.run([sys.executable, "-c", DRIVER % dict(oracle=ORACLE)], env=san_env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0 and "sanitized runs ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
