"""ctypes binding of the CPU ORACLE (oracle/libmci_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (mcintegration.jl_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MCI_ORACLE_SO") or os.path.join(_HERE, "libmci_oracle.so")   # (MCI_ORACLE_SO: the sanitizer build, oracle/Makefile `sanitize`)

CONTINUOUS, DISCRETE, FERMIK = 0, 1, 2
VEGAS, VEGASMC, MCMC = 0, 1, 2
PROB_CREATE, PROB_SHIFT = 0, 1

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_long_p = C.POINTER(C.c_long)
INTEGRAND_FN = C.CFUNCTYPE(None, c_double_p, c_double_p, c_double_p)


def build(force=False):
    """gcc-build the oracle in place (recipe: oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("mci_oracle.c", "mci_oracle_integrands.c", "mci_oracle.h")]
    if os.environ.get("MCI_ORACLE_SO"):
        return _SO   # (built by whoever pointed here)
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _SO


class _Leaf(C.Structure):
    _fields_ = [("kind", C.c_int), ("pool", C.c_int), ("lower", C.c_double), ("upper", C.c_double),
                ("npts", C.c_int), ("nbin", C.c_int), ("alpha", C.c_double), ("adapt", C.c_int),
                ("grid", c_double_p), ("hist", c_double_p), ("accumulation", c_double_p),
                ("distribution", c_double_p), ("P", C.c_int), ("data", c_double_p), ("gidx", c_long_p),
                ("prob", c_double_p), ("width", C.c_int)]


class _Config(C.Structure):
    _fields_ = [("nleaf", C.c_int), ("npool", C.c_int), ("Ni", C.c_int), ("leaf", C.POINTER(_Leaf)),
                ("pool_leaf0", c_int_p), ("pool_nleaf", c_int_p), ("pool_offset", c_int_p),
                ("pool_prob", C.POINTER(c_double_p)), ("pool_prob_cache", c_double_p), ("dof", c_int_p),
                ("maxdof", c_int_p), ("ndraw", C.c_int), ("draw_leaf", c_int_p), ("draw_slot", c_int_p),
                ("nobs", C.c_int), ("obs_off", c_int_p), ("obs_nbin", c_int_p), ("obs_bin_draw", c_int_p),
                ("observable", c_double_p), ("normalization", C.c_double), ("neval", C.c_long),
                ("reweight", c_double_p), ("visited", c_double_p), ("propose", c_double_p),
                ("accept", c_double_p), ("prob_mode", C.c_int), ("npa", C.c_int), ("pam", C.c_int), ("rng_bits", C.c_int), ("nneighbor", c_int_p),
                ("neighbor", C.POINTER(c_int_p)), ("thermal_ratio", C.c_double), ("reweight_goal", c_double_p),
                ("ncomp", C.c_int), ("measure_fn", C.c_void_p), ("pool_width", c_int_p), ("draw_comp", c_int_p),
                ("hold_hist", C.POINTER(C.c_ulonglong)), ("carry", C.c_void_p), ("carry_owner", C.c_int),
                ("carry_load", C.c_int), ("carry_store", C.c_int), ("carry_lb", C.c_long)]


class _Result(C.Structure):
    _fields_ = [("niter", C.c_int), ("nobs", C.c_int), ("Ni", C.c_int), ("iter_mean", c_double_p),
                ("iter_std", c_double_p), ("mean", c_double_p), ("stdev", c_double_p), ("chi2", c_double_p),
                ("neval", C.c_long)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    L.mcio_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mcio_philox4x32_r.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int]
    L.mcio_set_rng_rounds.argtypes = [C.c_int]
    L.mcio_uniform.restype = C.c_double
    L.mcio_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]
    L.mcio_uniform32.restype = C.c_double
    L.mcio_uniform32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]
    L.mcio_locate.restype = C.c_long
    L.mcio_locate.argtypes = [c_double_p, C.c_long, C.c_double]
    L.mcio_smooth.argtypes = [c_double_p, C.c_long, C.c_double, c_double_p]
    L.mcio_rescale.argtypes = [c_double_p, C.c_long, C.c_double]
    L.mcio_train_continuous.argtypes = [c_double_p, C.c_long, c_double_p, C.c_double]
    L.mcio_train_discrete.argtypes = [c_double_p, C.c_long, C.c_double, c_double_p, c_double_p]
    L.mcio_config_create.restype = C.POINTER(_Config)
    L.mcio_config_create.argtypes = [C.c_int, c_int_p, c_int_p, c_double_p, c_double_p, c_int_p, c_double_p,
                                     c_int_p, c_int_p, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p]
    L.mcio_config_destroy.argtypes = [C.POINTER(_Config)]
    L.mcio_config_clone.restype = C.POINTER(_Config)
    L.mcio_config_clone.argtypes = [C.POINTER(_Config)]
    L.mcio_set_grid.argtypes = [C.POINTER(_Config), C.c_int, c_double_p, C.c_int]
    L.mcio_set_distribution.argtypes = [C.POINTER(_Config), C.c_int, c_double_p]
    L.mcio_maxdof.argtypes = [c_int_p, C.c_int, C.c_int, c_int_p]
    L.mcio_clear_statistics.argtypes = [C.POINTER(_Config)]
    L.mcio_add_config.argtypes = [C.POINTER(_Config), C.POINTER(_Config)]
    L.mcio_train.argtypes = [C.POINTER(_Config)]
    for name in ("mcio_create", "mcio_shift"):
        fn = getattr(L, name)
        fn.restype = C.c_double
        fn.argtypes = [C.POINTER(_Config), C.c_int, C.c_int, C.c_double]
    L.mcio_shift_rollback.argtypes = [C.POINTER(_Config), C.c_int, C.c_int]
    for name in ("mcio_pool_shift", "mcio_pool_create"):
        fn = getattr(L, name)
        fn.restype = C.c_double
        fn.argtypes = [C.POINTER(_Config), C.c_int, C.c_int, c_double_p]
    L.mcio_pool_shift_rollback.argtypes = [C.POINTER(_Config), C.c_int, C.c_int]
    L.mcio_total_probability.restype = C.c_double
    L.mcio_total_probability.argtypes = [C.POINTER(_Config)]
    for name in ("mcio_probability", "mcio_padding_probability"):
        fn = getattr(L, name)
        fn.restype = C.c_double
        fn.argtypes = [C.POINTER(_Config), C.c_int]
    L.mcio_vegas_block.argtypes = [C.POINTER(_Config), C.c_void_p, c_double_p, C.c_uint64, C.c_uint32, C.c_long,
                                   C.c_long, C.c_long]
    L.mcio_vegasmc_block.argtypes = [C.POINTER(_Config), C.c_void_p, c_double_p, C.c_uint64, C.c_uint32, C.c_long,
                                     C.c_long, C.c_long, C.c_long]
    L.mcio_mcmc_block.argtypes = [C.POINTER(_Config), C.c_void_p, c_double_p, C.c_uint64, C.c_uint32, C.c_long,
                                  C.c_long, C.c_long, C.c_long]
    L.mcio_mcmc_burnin.restype = C.c_long
    L.mcio_mcmc_burnin.argtypes = [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_double]
    L.mcio_set_neighbor.argtypes = [C.POINTER(_Config), c_int_p, c_int_p]
    L.mcio_set_thermal_ratio.argtypes = [C.POINTER(_Config), C.c_double]
    L.mcio_set_chain_carry.argtypes = [C.POINTER(_Config), C.c_int]
    L.mcio_set_reweight_goal.argtypes = [C.POINTER(_Config), c_double_p]
    L.mcio_set_ncomp.argtypes = [C.POINTER(_Config), C.c_int]
    L.mcio_set_measure.argtypes = [C.POINTER(_Config), C.c_void_p]
    L.mcio_pool_remove.restype = C.c_double
    L.mcio_pool_remove.argtypes = [C.POINTER(_Config), C.c_int, C.c_int]
    L.mcio_pool_swap.restype = C.c_double
    L.mcio_pool_swap.argtypes = [C.POINTER(_Config), C.c_int, C.c_int, C.c_int]
    L.mcio_standardize_block.argtypes = [C.c_long, C.c_long, C.c_long, c_long_p, c_long_p]
    L.mcio_mean_std.argtypes = [c_double_p, c_double_p, C.c_long, C.c_long, c_double_p, c_double_p]
    L.mcio_average.argtypes = [c_double_p, c_double_p, C.c_long, C.c_long, C.c_long, C.c_long, c_double_p,
                               c_double_p, c_double_p]
    L.mcio_do_reweight.argtypes = [c_double_p, c_double_p, C.c_long, C.c_double, c_double_p]
    L.mcio_resample_chains.argtypes = [C.POINTER(C.c_int), C.c_long, C.c_int, c_double_p, c_double_p, C.c_long, C.POINTER(C.c_long)]
    L.mcio_resample_chains.restype = None
    L.mcio_resample_weighted.argtypes = [c_double_p, C.c_long, C.c_long, C.POINTER(C.c_long)]
    L.mcio_resample_weighted.restype = None
    L.mcio_integrate.argtypes = [C.POINTER(_Config), C.c_int, C.c_void_p, c_double_p, C.c_long, C.c_int, C.c_long,
                                 C.c_int, C.c_int, C.c_double, C.c_long, C.c_uint64, C.c_int, C.c_long,
                                 C.POINTER(_Result)]
    L.mcio_iteration.argtypes = [C.POINTER(_Config), C.c_int, C.c_void_p, c_double_p, C.c_long, C.c_long, C.c_long,
                                 C.c_uint32, C.c_long, C.c_uint64, C.c_int, C.c_long, c_double_p]
    L.mcio_packed_size.restype = C.c_long
    L.mcio_packed_size.argtypes = [C.POINTER(_Config)]
    L.mcio_result_create.restype = C.POINTER(_Result)
    L.mcio_result_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.mcio_result_destroy.argtypes = [C.POINTER(_Result)]
    L.mcio_builtin.restype = C.c_void_p
    L.mcio_builtin.argtypes = [C.c_char_p]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def philox(ctr, key, rounds=10):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().mcio_philox4x32_r(c, k, o, int(rounds))
    return [int(v) for v in o]


def set_rng_rounds(rounds):
    """Philox4x32 rounds of every stream of this process: 10 (default) or 7 (mirror of mci_set_rng_rounds)"""
    assert rounds in (10, 7)
    lib().mcio_set_rng_rounds(int(rounds))


def uniform(seed, stream, index, k, bits=52):
    """uniform k of (stream, index): 52 random mantissa bits (default) or the opt-in 32-bit stream of :vegas"""
    return lib().mcio_uniform32(seed, stream, index, k) if bits == 32 else lib().mcio_uniform(seed, stream, index, k)


def locate(acc, p):
    a = np.ascontiguousarray(acc, dtype=np.float64)
    return int(lib().mcio_locate(_dp(a), len(a), float(p)))


def smooth(dist, factor=6.0):
    d = np.ascontiguousarray(dist, dtype=np.float64)
    out = np.empty_like(d)
    lib().mcio_smooth(_dp(d), len(d), float(factor), _dp(out))
    return out


def rescale(dist, alpha=1.5):
    d = np.array(dist, dtype=np.float64)
    rc = lib().mcio_rescale(_dp(d), len(d), float(alpha))
    if rc:
        raise AssertionError("rescale assertion %d" % rc)
    return d


def train_continuous(grid, hist, alpha):
    g = np.array(grid, dtype=np.float64)
    h = np.array(hist, dtype=np.float64)
    rc = lib().mcio_train_continuous(_dp(g), len(g), _dp(h), float(alpha))
    if rc:
        raise AssertionError("train! assertion %d" % rc)
    return g


def train_discrete(hist, alpha):
    h = np.array(hist, dtype=np.float64)
    dist = np.empty(len(h))
    acc = np.empty(len(h) + 1)
    rc = lib().mcio_train_discrete(_dp(h), len(h), float(alpha), _dp(dist), _dp(acc))
    if rc:
        raise AssertionError("train! assertion %d" % rc)
    return dist, acc


def maxdof(dof):
    d = np.ascontiguousarray(dof, dtype=np.int32)
    out = np.zeros(d.shape[1], dtype=np.int32)
    lib().mcio_maxdof(_ip(d), d.shape[0], d.shape[1], _ip(out))
    return out


def standardize_block(neval, nblock, nworker=1):
    a, b = C.c_long(), C.c_long()
    lib().mcio_standardize_block(int(neval), int(nblock), int(nworker), C.byref(a), C.byref(b))
    return a.value, b.value


def mean_std(obs_sum, obs_sq, block):
    s = np.ascontiguousarray(obs_sum, dtype=np.float64)
    q = np.ascontiguousarray(obs_sq, dtype=np.float64)
    m, e = np.empty_like(s), np.empty_like(s)
    lib().mcio_mean_std(_dp(s), _dp(q), len(s), int(block), _dp(m), _dp(e))
    return m, e


def average(iter_mean, iter_std, init=1, max=None):
    m = np.ascontiguousarray(iter_mean, dtype=np.float64)
    e = np.ascontiguousarray(iter_std, dtype=np.float64)
    if max is None:
        max = len(m)
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().mcio_average(_dp(m), _dp(e), len(m), 1, int(init), int(max), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def do_reweight(reweight, visited, gamma=1.0, goal=None):
    r = np.array(reweight, dtype=np.float64)
    v = np.ascontiguousarray(visited, dtype=np.float64)
    g = None if goal is None else np.ascontiguousarray(goal, dtype=np.float64)
    lib().mcio_do_reweight(_dp(r), _dp(v), len(r), float(gamma), None if g is None else _dp(g))
    return r


def resample_chains(curr_old, rw_now, rw_used, n_new):
    """which stored chain every chain of the next :mcmc launch continues (mcio_resample_chains; mirror of k_resample_chains)"""
    co = np.ascontiguousarray(curr_old, dtype=np.int32)
    a, b = np.ascontiguousarray(rw_now, dtype=np.float64), np.ascontiguousarray(rw_used, dtype=np.float64)
    src = (C.c_long * int(n_new))()
    lib().mcio_resample_chains(co.ctypes.data_as(C.POINTER(C.c_int)), len(co), len(a), _dp(a), _dp(b), int(n_new), src)
    return np.array(list(src), dtype=np.int64)


def resample_weighted(w, n_new):
    """which stored chain every chain of the next :vegasmc launch continues, given new target / old target per stored chain
    (mcio_resample_weighted; mirror of k_resample_chains' w_chain path)"""
    a = np.ascontiguousarray(w, dtype=np.float64)
    src = (C.c_long * int(n_new))()
    lib().mcio_resample_weighted(_dp(a), len(a), int(n_new), src)
    return np.array(list(src), dtype=np.int64)


def builtin(name):
    p = lib().mcio_builtin(name.encode())
    if not p:
        raise KeyError(name)
    return p


class Config:
    """Oracle Configuration (ref: src/configuration.jl:105-194).

    leaves: list of dicts {kind, pool, lower, upper, npts, alpha, adapt}; dof: [Ni][npool].
    """

    def __init__(self, leaves, dof, pool_offset=None, obs_nbin=None, obs_bin_draw=None, prob_mode=PROB_CREATE):
        L = lib()
        n = len(leaves)
        dof = np.ascontiguousarray(dof, dtype=np.int32)
        Ni, npool = dof.shape
        kind = np.array([lf["kind"] for lf in leaves], dtype=np.int32)
        pool = np.array([lf["pool"] for lf in leaves], dtype=np.int32)
        lower = np.array([lf["lower"] for lf in leaves], dtype=np.float64)
        upper = np.array([lf["upper"] for lf in leaves], dtype=np.float64)
        npts = np.array([lf.get("npts", 1000) for lf in leaves], dtype=np.int32)
        alpha = np.array([lf.get("alpha", 2.0) for lf in leaves], dtype=np.float64)
        adapt = np.array([1 if lf.get("adapt", True) else 0 for lf in leaves], dtype=np.int32)
        off = np.zeros(npool, dtype=np.int32) if pool_offset is None else np.ascontiguousarray(pool_offset, dtype=np.int32)
        onb = None if obs_nbin is None else np.ascontiguousarray(obs_nbin, dtype=np.int32)
        obd = None if obs_bin_draw is None else np.ascontiguousarray(obs_bin_draw, dtype=np.int32)
        self.p = L.mcio_config_create(n, _ip(kind), _ip(pool), _dp(lower), _dp(upper), _ip(npts), _dp(alpha),
                                      _ip(adapt), _ip(off), npool, Ni, _ip(dof),
                                      None if onb is None else _ip(onb), None if obd is None else _ip(obd))
        self.p.contents.prob_mode = prob_mode
        for i, lf in enumerate(leaves):
            if lf.get("grid") is not None:
                g = np.ascontiguousarray(lf["grid"], dtype=np.float64)
                L.mcio_set_grid(self.p, i, _dp(g), len(g))
            if lf.get("distribution") is not None:
                d = np.ascontiguousarray(lf["distribution"], dtype=np.float64)
                L.mcio_set_distribution(self.p, i, _dp(d))

    def __del__(self):
        try:
            lib().mcio_config_destroy(self.p)
        except Exception:
            pass

    # --- views -------------------------------------------------------------------------------
    @property
    def c(self):
        return self.p.contents

    def leaf(self, i):
        return self.c.leaf[i]

    def grid(self, i):
        lf = self.leaf(i)
        return np.ctypeslib.as_array(lf.grid, shape=(lf.npts,)).copy()

    def set_grid(self, i, grid):
        g = np.ascontiguousarray(grid, dtype=np.float64)
        lib().mcio_set_grid(self.p, i, _dp(g), len(g))

    def hist(self, i):
        lf = self.leaf(i)
        return np.ctypeslib.as_array(lf.hist, shape=(lf.nbin,)).copy()

    def distribution(self, i):
        lf = self.leaf(i)
        return np.ctypeslib.as_array(lf.distribution, shape=(lf.nbin,)).copy()

    def accumulation(self, i):
        lf = self.leaf(i)
        return np.ctypeslib.as_array(lf.accumulation, shape=(lf.nbin + 1,)).copy()

    def pool_data(self, i):
        lf = self.leaf(i)
        return np.ctypeslib.as_array(lf.data, shape=(lf.P + 1,))[1:].copy()

    def pool_prob(self, i):
        lf = self.leaf(i)
        return np.ctypeslib.as_array(lf.prob, shape=(lf.P + 1,))[1:].copy()

    def pool_gidx(self, i):
        lf = self.leaf(i)
        return np.ctypeslib.as_array(lf.gidx, shape=(lf.P + 1,))[1:].copy()

    @property
    def observable(self):
        return np.ctypeslib.as_array(self.c.observable, shape=(self.c.nobs,)).copy()

    @property
    def reweight(self):
        return np.ctypeslib.as_array(self.c.reweight, shape=(self.c.Ni + 1,)).copy()

    @property
    def visited(self):
        return np.ctypeslib.as_array(self.c.visited, shape=(self.c.Ni + 1,)).copy()

    @property
    def hold_hist(self):
        """:mcmc chains of the last iteration by bit_width(longest holding time)"""
        return np.array([self.c.hold_hist[b] for b in range(64)], dtype=np.uint64)

    @property
    def ndraw(self):
        return self.c.ndraw

    @property
    def nobs(self):
        return self.c.nobs

    def packed_size(self):
        return int(lib().mcio_packed_size(self.p))

    # --- operations --------------------------------------------------------------------------
    def create(self, leaf, idx, u):
        return lib().mcio_create(self.p, leaf, idx, float(u))

    def shift(self, leaf, idx, u):
        return lib().mcio_shift(self.p, leaf, idx, float(u))

    def pool_shift(self, vi, idx, us):
        u = np.ascontiguousarray(us, dtype=np.float64)
        return lib().mcio_pool_shift(self.p, vi, idx, _dp(u))

    def pool_create(self, vi, idx, us):
        u = np.ascontiguousarray(us, dtype=np.float64)
        return lib().mcio_pool_create(self.p, vi, idx, _dp(u))

    def total_probability(self):
        return lib().mcio_total_probability(self.p)

    def probability(self, i):
        return lib().mcio_probability(self.p, i)

    def padding_probability(self, i):
        return lib().mcio_padding_probability(self.p, i)

    def clear_statistics(self):
        lib().mcio_clear_statistics(self.p)

    def train(self):
        lib().mcio_train(self.p)

    def vegas_block(self, f, ud, seed, iteration, block_index, neval, measurefreq=1):
        u = np.ascontiguousarray(ud if ud is not None else [0.0], dtype=np.float64)
        return lib().mcio_vegas_block(self.p, _fnptr(f), _dp(u), seed, iteration, block_index, neval, measurefreq)

    def vegasmc_block(self, f, ud, seed, iteration, block_index, neval, measurefreq=1, nchain=1):
        u = np.ascontiguousarray(ud if ud is not None else [0.0], dtype=np.float64)
        return lib().mcio_vegasmc_block(self.p, _fnptr(f), _dp(u), seed, iteration, block_index, neval,
                                        measurefreq, nchain)

    def mcmc_block(self, f, ud, seed, iteration, block_index, neval, measurefreq=1, nchain=1):
        u = np.ascontiguousarray(ud if ud is not None else [0.0], dtype=np.float64)
        return lib().mcio_mcmc_block(self.p, _fnptr(f), _dp(u), seed, iteration, block_index, neval, measurefreq, nchain)

    def set_neighbor(self, neighbor):
        """neighbor: list of lists of 0-based integrand indices (index Ni = normalisation)"""
        off = np.zeros(len(neighbor) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(n) for n in neighbor])
        flat = np.ascontiguousarray([j for n in neighbor for j in n], dtype=np.int32)
        if lib().mcio_set_neighbor(self.p, _ip(off), _ip(flat)):
            raise ValueError("bad neighbor lists")

    def neighbor(self):
        return [[self.c.neighbor[d][j] for j in range(self.c.nneighbor[d])] for d in range(self.c.Ni + 1)]

    def set_ncomp(self, ncomp):
        """2 = ComplexF64 weights stored as (re, im); obs_nbin must already count doubles"""
        lib().mcio_set_ncomp(self.p, int(ncomp))

    def set_measure(self, fnptr):
        """raw pointer from compile_c_measure (or None for the default measure)"""
        lib().mcio_set_measure(self.p, fnptr)

    def set_chain_carry(self, mode):
        """mirror of mci_set_chain_carry: "auto" / -1 (default) and "on" / 1 carry both chain solvers, "off" / 0 none"""
        lib().mcio_set_chain_carry(self.p, {"auto": -1, "off": 0, "on": 1, -1: -1, 0: 0, 1: 1}[mode])

    def set_reweight(self, r):
        """config.reweight (configuration.jl:50), e.g. after do_reweight"""
        r = np.ascontiguousarray(r, dtype=np.float64)
        assert len(r) == self.c.Ni + 1
        for i, v in enumerate(r):
            self.c.reweight[i] = float(v)

    def set_thermal_ratio(self, r):
        lib().mcio_set_thermal_ratio(self.p, float(r))

    def set_rng_bits(self, bits):
        """:vegas sample stream: 52 or 32 random bits per draw (mirror of mci_set_rng_bits)"""
        assert bits in (52, 32)
        self.c.rng_bits = int(bits)

    def set_reweight_goal(self, goal):
        if goal is None:
            lib().mcio_set_reweight_goal(self.p, None)
        else:
            g = np.ascontiguousarray(goal, dtype=np.float64)
            assert len(g) == self.c.Ni + 1
            lib().mcio_set_reweight_goal(self.p, _dp(g))

    def pool_remove(self, vi, idx):
        return lib().mcio_pool_remove(self.p, vi, idx)

    def pool_swap(self, vi, idx1, idx2):
        return lib().mcio_pool_swap(self.p, vi, idx1, idx2)

    def iteration(self, solver, f, ud, nevalperblock, block_lo, block_hi, iteration, seed, measurefreq=1,
                  nthreads=1, nchain=1):
        """Run blocks [block_lo, block_hi) of one iteration; returns the packed statistics buffer
        [obsSum | obsSqSum | normalization | neval | visited | histograms] (no training)."""
        u = np.ascontiguousarray(ud if ud is not None else [0.0], dtype=np.float64)
        out = np.zeros(self.packed_size())
        rc = lib().mcio_iteration(self.p, solver, _fnptr(f), _dp(u), int(nevalperblock), int(block_lo), int(block_hi),
                                  int(iteration), int(measurefreq), int(seed), int(nthreads), int(nchain), _dp(out))
        if rc:
            raise RuntimeError("oracle iteration failed (%d)" % rc)
        return out

    def integrate(self, solver, f, ud, neval, niter=10, block=16, ignore=None, adapt=True, gamma=1.0,
                  measurefreq=1, seed=1234, nthreads=1, nchain=1):
        if ignore is None:
            ignore = 1 if adapt else 0
        u = np.ascontiguousarray(ud if ud is not None else [0.0], dtype=np.float64)
        r = lib().mcio_result_create(niter, self.c.nobs, self.c.Ni)
        rc = lib().mcio_integrate(self.p, solver, _fnptr(f), _dp(u), int(neval), niter, int(block), ignore,
                                  1 if adapt else 0, float(gamma), int(measurefreq), int(seed), int(nthreads),
                                  int(nchain), r)
        rr = r.contents
        n = rr.nobs
        res = dict(
            rc=rc,
            mean=np.ctypeslib.as_array(rr.mean, shape=(n,)).copy(),
            stdev=np.ctypeslib.as_array(rr.stdev, shape=(n,)).copy(),
            chi2=np.ctypeslib.as_array(rr.chi2, shape=(n,)).copy(),
            iter_mean=np.ctypeslib.as_array(rr.iter_mean, shape=(niter, n)).copy(),
            iter_std=np.ctypeslib.as_array(rr.iter_std, shape=(niter, n)).copy(),
            neval=int(rr.neval),
        )
        lib().mcio_result_destroy(r)
        return res


_keepalive = []


def _fnptr(f):
    """f: builtin name, raw pointer, or python callable (x: ndarray, ud: ndarray) -> sequence."""
    if isinstance(f, str):
        return builtin(f)
    if isinstance(f, int):
        return f
    raise TypeError("integrand must be a builtin name or a C function pointer")


def compile_c_measure(body):
    """gcc-compile a measure body (the same text the HIP path JIT-compiles): in scope `x`, `rw`, `ud`, `idx` and
    `obs_add(k, v)` which accumulates v into flat observable k."""
    import hashlib
    import tempfile
    src = ("#include <math.h>\n#define obs_add(k, v) (obs[(k)] += (v))\n"
           "void mci_user_measure(const double* x, const double* rw, const double* ud, int idx, double* obs) {\n"
           "(void)x; (void)rw; (void)ud; (void)idx;\n%s\n}\n" % body)
    h = hashlib.sha1(src.encode()).hexdigest()[:16]
    d = os.path.join(tempfile.gettempdir(), "mci_oracle_user")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "m_%s.so" % h)
    if not os.path.exists(so):
        cfile = os.path.join(d, "m_%s.c" % h)
        with open(cfile, "w") as fh:
            fh.write(src)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, cfile, "-lm"])
    L = C.CDLL(so)
    _keepalive.append(L)
    return C.cast(L.mci_user_measure, C.c_void_p).value


def compile_c_integrand(body, ni=1, name="user"):
    """gcc-compile a C integrand body (same text that the HIP path JIT-compiles) for the oracle.

    body sees `x`, `w`, `ud` like the device snippet.  Returns a raw function pointer.
    """
    import hashlib
    import tempfile
    src = ("#include <math.h>\n#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n"
           "void mci_user_integrand(const double* x, double* w, const double* ud) {\n%s\n}\n" % body)
    h = hashlib.sha1(src.encode()).hexdigest()[:16]
    d = os.path.join(tempfile.gettempdir(), "mci_oracle_user")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "u_%s.so" % h)
    if not os.path.exists(so):
        cfile = os.path.join(d, "u_%s.c" % h)
        with open(cfile, "w") as fh:
            fh.write(src)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, cfile, "-lm"])
    L = C.CDLL(so)
    _keepalive.append(L)
    return C.cast(L.mci_user_integrand, C.c_void_p).value
