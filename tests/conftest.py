import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    # tests that keep a per-sample closure on the host on purpose (trace=False) hear the engine's one-time "correct, but slow" note;
    # the tests ABOUT that note record it themselves (pytest.warns / catch_warnings override this filter)
    config.addinivalue_line("filterwarnings", "ignore:the host (integrand|measure) closure raised on a batch:RuntimeWarning")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "mcintegration.jl_amd", "lib", "libmci_hip.so")
    demo = os.path.join(ROOT, "examples", "mci_demo")
    if not (os.path.exists(lib) and os.path.exists(demo)):
        import __graft_entry__
        __graft_entry__.build()
    # the suite compiles with the ROCm installation's hiprtc + comgr (the compiler build() pre-fills the kernel cache with), pinned before
    # any test module imports torch: PyTorch bundles another compiler build, and the first comgr loaded into a process serves everybody
    # (comgr + hiprtc only -- libmci_hip.so and the HIP runtime are loaded by the first test that needs them, after collection has imported
    # every test module and with it torch: one runtime per process, as before)
    import mcintegration_jl_amd
    mcintegration_jl_amd.use_rocm_compiler()


@pytest.fixture(scope="session")
def oracle():
    import mci_oracle
    mci_oracle.build()
    return mci_oracle


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        return json.load(fh)


class _Overrides:
    """layout decisions of mci_problem_create forced for a test (csrc/mci_debug.h mci_debug_override): process-wide, consulted by the
    next Engine(...); everything set through the fixture is taken back when the test ends"""

    def __init__(self):
        self.keys = set()

    def set(self, key, value):
        from mcintegration_jl_amd._lib import check, lib
        check(lib().mci_debug_override(key.encode(), int(value), 1))
        self.keys.add(key)

    def clear(self, key=None):
        from mcintegration_jl_amd._lib import check, lib
        for k in ([key] if key else list(self.keys)):
            check(lib().mci_debug_override(k.encode(), 0, 0))
            self.keys.discard(k)


@pytest.fixture
def overrides():
    o = _Overrides()
    yield o
    o.clear()


@pytest.fixture(autouse=True)
def _note_process_groups(request):
    """remember, for pytest_unconfigure, whether any test of this session initialised a torch.distributed process group"""
    yield
    td = sys.modules.get("torch.distributed")
    try:
        if td is not None and td.is_available() and td.is_initialized():
            request.config._mci_made_process_group = True
    except Exception:
        pass


def pytest_sessionfinish(session, exitstatus):
    """Release engines, streams and communicators while the HIP runtime and RCCL are fully alive (see
    mcintegration_jl_amd.engine.shutdown): the test processes create hundreds of problems and several communicators."""
    import gc
    gc.collect()
    mod = sys.modules.get("mcintegration_jl_amd")
    if mod is not None:
        mod.shutdown()
    session.config._mci_exitstatus = int(exitstatus)


@pytest.hookimpl(trylast=True)
def pytest_unconfigure(config):
    """A process that created BOTH the library's own RCCL communicator and a torch NCCL process group (tests/test_hip_battery.py) can
    abort inside the C++ static destructors of RCCL / the HIP runtime after the interpreter is gone -- glibc "double free or corruption",
    exit code 134 with every test passed (seen with exactly those tests selected, also on the code this round started from; outside this repository's
    code: everything of ours has been released by then).  Like bench.py, such a process leaves through os._exit once pytest has
    reported, with pytest's own exit status.  ONLY such a process: one that merely imported torch keeps the normal interpreter and
    library teardown, so that an abort in OUR static destructors, atexit handlers or orphaned compile threads shows as exit code 134."""
    td = sys.modules.get("torch.distributed")
    made_group = False
    try:
        made_group = bool(td is not None and (getattr(config, "_mci_made_process_group", False) or os.environ.get("MCI_TEST_MADE_PROCESS_GROUP")
                                              or (td.is_available() and td.is_initialized())))
    except Exception:
        made_group = False
    if made_group and hasattr(config, "_mci_exitstatus"):
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(config._mci_exitstatus)
