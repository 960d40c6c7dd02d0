"""ctypes binding of include/mci.h.  The library is built in-tree by __graft_entry__.build();
there is no Python/CPU fallback: if it is missing, importing the engine fails loudly."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_PKG, "lib", "libmci_hip.so")

MCI_OK = 0
ERR_NAMES = {1: "INVALID", 2: "HIP", 3: "COMPILE", 4: "NORMALIZATION", 5: "HISTOGRAM", 6: "COMM", 7: "NO_DEVICE"}
CONTINUOUS, DISCRETE, FERMIK = 0, 1, 2
VEGAS, VEGASMC, MCMC = 0, 1, 2
ABI_VERSION = 5   # include/mci.h MCI_ABI_VERSION
SOLVERS = {"vegas": VEGAS, "vegasmc": VEGASMC, "mcmc": MCMC, VEGAS: VEGAS, VEGASMC: VEGASMC, MCMC: MCMC}

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class MCIError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("[MCI_ERR_%s] %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


class LeafDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pool", C.c_int32), ("lower", C.c_double), ("upper", C.c_double),
                ("npoints", C.c_int32), ("alpha", C.c_double), ("adapt", C.c_int32), ("init", c_double_p)]


class ProblemDesc(C.Structure):
    _fields_ = [("nleaf", C.c_int32), ("leaves", C.POINTER(LeafDesc)), ("npool", C.c_int32),
                ("nintegrand", C.c_int32), ("dof", c_int32_p), ("obs_nbin", c_int32_p), ("obs_bin_draw", c_int32_p),
                ("neighbor_offsets", c_int32_p), ("neighbor_list", c_int32_p), ("ncomp", C.c_int32)]


class IntegrateArgs(C.Structure):
    _fields_ = [("solver", C.c_int32), ("neval", C.c_int64), ("niter", C.c_int32), ("block", C.c_int64),
                ("ignore", C.c_int32), ("adapt", C.c_int32), ("gamma", C.c_double), ("measurefreq", C.c_int64),
                ("seed", C.c_uint64), ("nchain", C.c_int64), ("first_iteration", C.c_int32),
                ("thermal_ratio", C.c_double), ("reweight_goal", c_double_p)]


HOST_INTEGRAND_FN = C.CFUNCTYPE(C.c_int, c_double_p, c_double_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)
HOST_INTEGRAND_IDX_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int32), c_double_p, c_double_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)
HOST_MEASURE_FN = C.CFUNCTYPE(C.c_int, c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, c_double_p, C.c_int32, C.c_void_p)
HOST_MEASURE_IDX_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int32), c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, c_double_p,
                                  C.c_int32, C.c_void_p)


class ResultC(C.Structure):
    _fields_ = [("niter", C.c_int32), ("nobs", C.c_int32), ("iter_mean", c_double_p), ("iter_std", c_double_p),
                ("mean", c_double_p), ("stdev", c_double_p), ("chi2", c_double_p), ("neval", C.c_int64),
                ("seconds", C.c_double), ("visited", c_double_p), ("correlated", C.c_int32), ("warmup", C.c_int32)]


# every symbol include/mci.h declares: (name, restype, argtypes); DEBUG_SIGNATURES: the test hooks of csrc/mci_debug.h
_VP = C.c_void_p
SIGNATURES = [
    ("mci_ctx_create", C.c_int, [C.c_int32, C.POINTER(_VP)]),
    ("mci_ctx_destroy", C.c_int, [_VP]),
    ("mci_last_error", C.c_char_p, []),
    ("mci_device_count", C.c_int, [c_int32_p]),
    ("mci_ctx_stream", _VP, [_VP]),
    ("mci_comm_unique_id", C.c_int, [_VP]),
    ("mci_comm_init", C.c_int, [_VP, C.c_int32, C.c_int32, _VP]),
    ("mci_comm_rank", C.c_int, [_VP, c_int32_p, c_int32_p]),
    ("mci_comm_sum", C.c_int, [_VP, c_double_p, C.c_int32]),
    ("mci_problem_create", C.c_int, [_VP, C.POINTER(ProblemDesc), C.POINTER(_VP)]),
    ("mci_problem_destroy", C.c_int, [_VP]),
    ("mci_set_integrand_source", C.c_int, [_VP, C.c_char_p, c_double_p, C.c_int32]),
    ("mci_set_integrand_host", C.c_int, [_VP, _VP, _VP]),
    ("mci_set_integrand_host_indexed", C.c_int, [_VP, _VP, _VP]),
    ("mci_set_measure_source", C.c_int, [_VP, C.c_char_p]),
    ("mci_set_measure_host", C.c_int, [_VP, _VP, _VP]),
    ("mci_set_measure_host_indexed", C.c_int, [_VP, _VP, _VP]),
    ("mci_compile", C.c_int, [_VP]),
    ("mci_compile_solver", C.c_int, [_VP, C.c_int32]),
    ("mci_kernel_code_object", C.c_int, [_VP, C.c_int32, C.c_char_p, C.c_int32]),
    ("mci_check_status", C.c_int, [_VP]),
    ("mci_set_launch", C.c_int, [_VP, C.c_int32, C.c_int32]),
    ("mci_problem_info", C.c_int, [_VP, c_int32_p, c_int32_p, C.POINTER(C.c_int64), c_int32_p, C.POINTER(C.c_int64)]),
    ("mci_get_histogram_copies", C.c_int, [_VP, c_int32_p]),
    ("mci_set_kernel_timing", C.c_int, [_VP, C.c_int32]),
    ("mci_iteration_run", C.c_int, [_VP, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_double]),
    ("mci_iteration_reduce", C.c_int, [_VP]),
    ("mci_iteration_finish", C.c_int, [_VP, C.c_int32, C.c_int64, C.c_int32, C.c_double, c_double_p, c_double_p]),
    ("mci_integrate", C.c_int, [_VP, C.POINTER(IntegrateArgs), C.POINTER(ResultC)]),
    ("mci_get_iteration_log", C.c_int, [_VP, C.c_int32, c_double_p]),
    ("mci_reserve_iteration_log", C.c_int, [_VP, C.c_int32]),
    ("mci_get_packed", C.c_int, [_VP, c_double_p, C.c_int64]),
    ("mci_set_packed", C.c_int, [_VP, c_double_p, C.c_int64]),
    ("mci_packed_device_ptr", _VP, [_VP]),
    ("mci_get_grid", C.c_int, [_VP, C.c_int32, c_double_p, C.c_int32]),
    ("mci_set_grid", C.c_int, [_VP, C.c_int32, c_double_p, C.c_int32]),
    ("mci_get_distribution", C.c_int, [_VP, C.c_int32, c_double_p, c_double_p, C.c_int32]),
    ("mci_set_distribution", C.c_int, [_VP, C.c_int32, c_double_p, C.c_int32]),
    ("mci_get_reweight", C.c_int, [_VP, c_double_p, C.c_int32]),
    ("mci_set_reweight", C.c_int, [_VP, c_double_p, C.c_int32]),
    ("mci_set_reweight_goal", C.c_int, [_VP, c_double_p, C.c_int32]),
    ("mci_get_acceptance", C.c_int, [_VP, c_double_p, c_double_p, C.c_int32]),
    ("mci_save_state", C.c_int, [_VP, C.c_char_p]),
    ("mci_load_state", C.c_int, [_VP, C.c_char_p]),
    ("mci_set_train_walk", C.c_int, [_VP, C.c_int32]),
    ("mci_set_rng_bits", C.c_int, [_VP, C.c_int32]),
    ("mci_set_rng_rounds", C.c_int, [_VP, C.c_int32]),
    ("mci_set_deterministic", C.c_int, [_VP, C.c_int32]),
    ("mci_set_chain_carry", C.c_int, [_VP, C.c_int32]),
    ("mci_set_iteration_counted", C.c_int, [_VP, C.c_int32]),
    ("mci_set_persistent", C.c_int, [_VP, C.c_int32]),
    ("mci_last_integrate_persistent", C.c_int, [_VP, C.POINTER(C.c_int32)]),
    ("mci_last_chain_launch", C.c_int, [_VP, C.POINTER(C.c_int64), c_int32_p]),
    ("mci_set_chain_speculation", C.c_int, [_VP, C.c_int32, C.c_double, C.c_int32]),
    ("mci_last_chain_speculation", C.c_int, [_VP, c_int32_p, c_int32_p]),
    ("mci_chain_speculation_status", C.c_int, [_VP, C.c_int32, c_int32_p]),
    ("mci_last_integrate_discarded", C.c_int, [_VP, C.POINTER(C.c_int64), c_int32_p]),
    ("mci_speculation_tree", C.c_int, [C.c_int32, C.c_double, C.c_int32, c_int32_p, c_int32_p, c_int32_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("mci_compile_chain_speculation", C.c_int, [_VP, C.c_int32]),
    ("mci_train", C.c_int, [_VP]),
    ("mci_sample_dump", C.c_int, [_VP, C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, c_double_p, c_double_p, c_double_p]),
    ("mci_kernel_times_ms", C.c_int, [_VP, C.POINTER(C.c_float), C.c_int32, c_int32_p, c_int32_p, c_int32_p]),
    ("mci_kernel_clocks", C.c_int, [_VP, c_double_p, C.c_int32, c_int32_p]),
    ("mci_comm_times_ms", C.c_int, [_VP, C.POINTER(C.c_float), C.c_int32, c_int32_p]),
    ("mci_comm_collectives", C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mci_reduce_size", C.c_int, [_VP, C.POINTER(C.c_int64)]),
    ("mci_external_reduce_done", C.c_int, [_VP]),
    ("mci_standardize_block", None, [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mci_chain_burnin", C.c_double, [C.c_int64, C.c_int64, C.c_int32]),
    ("mci_mcmc_burnin", C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    ("mci_mcmc_auto_chains", C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32]),
    ("mci_get_hold_histogram", C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    ("mci_mcmc_launch_valid", C.c_int, [_VP, c_int32_p, c_int32_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mci_iteration_discard", C.c_int, [_VP]),
    ("mci_get_block_means", C.c_int, [_VP, C.c_int32, c_double_p, C.POINTER(C.c_int64), c_int32_p]),
    ("mci_reset_block_log", C.c_int, [_VP]),
    ("mci_lineage_sums", None, [c_double_p, C.c_int64, C.c_int64, C.c_int64, c_double_p, C.c_int64, C.c_int64, c_double_p, c_double_p]),
    ("mci_maxdof", None, [c_int32_p, C.c_int32, C.c_int32, c_int32_p]),
    ("mci_mean_std", None, [c_double_p, c_double_p, C.c_int64, C.c_int64, c_double_p, c_double_p]),
    ("mci_average", None, [c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_int64, c_double_p, c_double_p, c_double_p]),
    ("mci_do_reweight", None, [c_double_p, c_double_p, C.c_int64, C.c_double, c_double_p]),
    ("mci_version", C.c_char_p, []),
    ("mci_abi_version", C.c_int32, []),
]
DEBUG_SIGNATURES = [
    ("mci_debug_persist_words", C.c_int, [_VP, C.POINTER(C.c_uint64), C.c_int32]),
    ("mci_debug_walk_counts", C.c_int, [_VP, C.POINTER(C.c_int64)]),
    ("mci_debug_plant_wrong_decision", C.c_int, [_VP, C.c_int32]),
    ("mci_debug_persist_spin_ticks", C.c_int, [_VP, C.c_uint64]),
    ("mci_debug_override", C.c_int, [C.c_char_p, C.c_int64, C.c_int32]),
    ("mci_debug_compiler_id", C.c_int, [C.c_char_p, C.c_char_p, C.c_int32]),
    ("mci_debug_split_chunks", C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mci_debug_mcmc_policy", C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
]

_lib = None


def library_path():
    return _SO


def use_rocm_compiler(rocm=None):
    """Pin the JIT's compiler to the ROCm installation's hiprtc + comgr (ROCM_PATH, default /opt/rocm) -- the toolchain the library itself
    was built with and its kernels are validated against.  hiprtc opens the code-object manager (libamd_comgr: the clang / LLVM that
    compiles) by soname, so the FIRST copy loaded into a process serves everybody; a process that imports PyTorch first gets the copies
    PyTorch bundles, another compiler build (the headline loop comes out 4 % longer and 2.2 % slower; the kernel cache is keyed by the
    compiler, so the two never mix).  Call this BEFORE `import torch` (bench.py, tests/conftest.py and __graft_entry__ do).  It loads the
    installation's comgr and hiprtc with RTLD_GLOBAL and NOTHING ELSE: not libmci_hip.so and with it no HIP runtime -- a process that goes
    on to import PyTorch keeps ONE runtime, PyTorch's, which libmci_hip.so then binds to as it always did (its streams and buffers, the
    library's RCCL communicator and PyTorch's all live in that one runtime); only the compiler is the installation's.  Returns True if
    it loaded them, False if ANOTHER comgr was in the process already (nothing is changed then; `compiler_id()` says which one compiles)."""
    root = rocm or os.environ.get("ROCM_PATH") or "/opt/rocm"
    mapped = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                path = line.split(None, 5)[-1].strip() if line.count(" ") >= 5 else ""
                if "amd_comgr" in path or "hiprtc" in path:
                    mapped.add(os.path.realpath(path))
    except OSError:
        pass

    def installed(names):
        for name in names:
            path = os.path.join(root, "lib", name)
            if os.path.exists(path):
                return path
        return None
    comgr, hiprtc = installed(("libamd_comgr.so.3", "libamd_comgr.so.2", "libamd_comgr.so")), installed(("libhiprtc.so.7", "libhiprtc.so"))
    have_comgr = [p for p in mapped if "amd_comgr" in p]
    if have_comgr and (comgr is None or os.path.realpath(comgr) not in have_comgr):
        return False                                       # another comgr serves this process already (PyTorch's, typically)
    # (a comgr that IS the installation's may be there before us -- rocprofv3's tool library maps it -- and then hiprtc still has to be pinned:
    # PyTorch's own libhiprtc in front of the installation's comgr is a third compiler identity, and a kernel-cache key of its own)
    done = False
    for path in (comgr, hiprtc):
        if path is not None:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            done = True
    return done


def compiler_id():
    """which compiler the library's JIT resolves to in this process: hiprtc version | hiprtc file | comgr file | target"""
    buf = C.create_string_buffer(1024)
    lib().mci_debug_compiler_id(None, buf, len(buf))
    return buf.value.decode()


def lib():
    """Load libmci_hip.so (fails loudly when it has not been built: no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError("%s is missing: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first; "
                              "the MI355X engine has no Python/CPU fallback" % _SO)
        L = C.CDLL(_SO, mode=C.RTLD_GLOBAL)
        for name, res, args in SIGNATURES + DEBUG_SIGNATURES:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.mci_abi_version() != ABI_VERSION:   # the ctypes structures below mirror ONE layout of include/mci.h
            raise ImportError("%s has ABI %d, this package was written against ABI %d: rebuild with `python __graft_entry__.py`" % (_SO, L.mci_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != MCI_OK:
        raise MCIError(rc, lib().mci_last_error().decode(errors="replace"))
