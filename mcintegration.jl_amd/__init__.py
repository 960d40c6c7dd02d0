"""Host-side mirror of MCIntegration.jl's public surface for the VEGAS / VegasMC path, driving the
MI355X engine (libmci_hip.so) through its C ABI (include/mci.h).

Same names and keyword meaning as the reference (src/MCIntegration.jl:20-47):
    integrate, Configuration, Continuous, Discrete, CompositeVar, Result, report, Dist
The integrand is HIP C++ source (a string or an `Integrand`) instead of a Julia closure: it is
JIT-compiled into the sample-batch kernel, exactly where Julia would inline the closure.
"""
from . import catalog  # noqa: F401
from ._lib import MCIError, compiler_id, lib, library_path, use_rocm_compiler  # noqa: F401
from .configuration import Configuration  # noqa: F401
from .engine import Engine, shutdown  # noqa: F401
from .integrand import HostIntegrand, HostMeasure, Integrand, Measure, bin_by  # noqa: F401
from .integrate import integrate, prefill_kernel_cache, standardize_block  # noqa: F401
from .solvers import MCMC, Vegas, VegasMC  # noqa: F401  (reference: modules Vegas, VegasMC, MCMC -- `Solver.montecarlo`, the seam of main.jl:253-264)
from .statistics import Result, average, mean_std, report  # noqa: F401
from . import trace  # noqa: F401
from .trace import TraceError, trace_integrand, trace_measure  # noqa: F401
from .variables import CompositeVar, Continuous, Discrete, FermiK  # noqa: F401
from . import variables as Dist  # noqa: F401  (reference: module Dist)



def disable_threading():
    """reference utility/parallel.jl:156-164 (asserts that Julia runs one thread and sets BLAS to one): the sample batch runs on the GPU
    and the host side is one thread per process, so there is nothing to switch off"""
    return None


__all__ = ["integrate", "Configuration", "Continuous", "Discrete", "CompositeVar", "FermiK", "Result", "report", "Vegas", "VegasMC", "MCMC",
           "Dist", "Engine", "Integrand", "HostIntegrand", "HostMeasure", "Measure", "bin_by", "catalog", "MCIError", "disable_threading"]
