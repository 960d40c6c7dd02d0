"""Engine: one mci_problem on one GPU (device-resident grids, histograms, statistics)."""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import LeafDesc, ProblemDesc, c_double_p, c_int32_p, check, lib
from .integrand import HostIntegrand, HostMeasure, Integrand, Measure
from .variables import ContinuousVar, FermiK

_ctx_cache = {}


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def context(device=0):
    """One mci_ctx (HIP stream + optional RCCL communicator) per device per process.
    device < 0 gives the offline, compile-only context (kernel cache pre-fill; no GPU needed)."""
    if device not in _ctx_cache:
        p = C.c_void_p()
        check(lib().mci_ctx_create(device, C.byref(p)))
        _ctx_cache[device] = p
    return _ctx_cache[device]


_live_engines = weakref.WeakSet()


def shutdown():
    """Close every live engine, then destroy every cached context (stream + RCCL communicator): the counterpart of
    MPI.Finalize.  Registered with atexit, so that device memory, streams and communicators are released while the HIP
    runtime and RCCL are still fully alive -- objects that survive into the interpreter's teardown are destroyed in an
    order nobody controls (seen: glibc "double free or corruption" at process exit, after all work had finished)."""
    for eng in list(_live_engines):
        try:
            eng.close()
        except Exception:
            pass
    for dev in list(_ctx_cache):
        try:
            lib().mci_ctx_destroy(_ctx_cache.pop(dev))
        except Exception:
            pass


atexit.register(shutdown)


def device_count():
    n = C.c_int32()
    lib().mci_device_count(C.byref(n))
    return n.value


class _HostObs(np.ndarray):
    """an observable handed to a host `measure` closure with a batch of records: `obs[i][bins] += w` with an integer ARRAY of bins
    (`obs[1][bin[1]] += weights[1]` of docs/src/index.md "Measure Histogram" over the records of a block) adds every record's weight to
    its bin -- numpy's own fancy `+=` would keep one of the records that share a bin.  An observable only accumulates
    (vegas/montecarlo.jl:156-161), so reading bins through an integer array gives zeros and storing through one adds."""

    def __new__(cls, n, dtype=float):
        return np.zeros(n, dtype=dtype).view(cls)

    @staticmethod
    def _bins(i):
        one = lambda q: isinstance(q, np.ndarray) and q.dtype.kind in "iu" and q.ndim >= 1
        return one(i) or (isinstance(i, tuple) and any(one(q) for q in i) and all(one(q) or isinstance(q, (int, np.integer)) for q in i))

    def __getitem__(self, i):
        if self._bins(i):
            return np.zeros(np.broadcast(*i).shape if isinstance(i, tuple) else i.shape, dtype=self.dtype)
        out = np.ndarray.__getitem__(self, i)
        return out.view(np.ndarray) if isinstance(out, np.ndarray) else out

    def __setitem__(self, i, v):
        if self._bins(i):
            np.add.at(self.view(np.ndarray), i, v)
            return
        np.ndarray.__setitem__(self, i, v)


def _slow_path_warning(what, unit, err):
    """said once per callback: a host closure that cannot take a batch is called per %(unit)s by the interpreter -- orders of magnitude
    below the traced path (integrate(..., trace=True) says why a closure was not traced)"""
    import warnings
    warnings.warn("the host %s closure raised on a batch of %ss (%s: %s) and is now called %s by %s: correct, but slow -- write it for "
                  "arrays, or make it traceable (integrate(..., trace=True) names what stops the tracer)"
                  % (what, unit, type(err).__name__, str(err)[:80], unit, unit), RuntimeWarning, stacklevel=2)


def _store_obs(O, obs, obs_nbin, nc):
    off = 0
    for o, nb in zip(obs, obs_nbin):
        o = np.asarray(o).reshape(-1)
        if nc == 2:
            O[off:off + nb:2], O[off + 1:off + nb:2] = o.real, o.imag
        else:
            O[off:off + nb] = o
        off += nb


class Engine:
    def __init__(self, config, integrand, measure=None, device=0, threads=None, wg_per_block=None, rng_bits=None, rng_rounds=None,
                 deterministic=False):
        L = lib()
        self.config = config
        self.device = device
        self.ctx = context(device)
        _live_engines.add(self)
        if isinstance(integrand, str):
            integrand = Integrand(integrand, config.userdata)
        elif callable(integrand) and not isinstance(integrand, (Integrand, HostIntegrand)):
            integrand = HostIntegrand(integrand)
        self.integrand = integrand
        if callable(measure) and not isinstance(measure, (Measure, HostMeasure)) and not hasattr(measure, "pool"):
            measure = HostMeasure(measure)   # a Python closure: host batch-callback path
        self.measure = measure               # None (the default measure), bin_by, Measure (device source) or HostMeasure
        leaves = config.leaves
        self._keep = []
        arr = (LeafDesc * len(leaves))()
        for i, lf in enumerate(leaves):
            d = arr[i]
            d.pool = config.leaf_pool[i]
            d.alpha, d.adapt = lf.alpha, 1 if lf.adapt else 0
            d.lower, d.upper = float(lf.lower), float(lf.upper)
            init = None
            # a variable that already lives in another, still open engine brings what it has learned there:
            # `integrate(...; var = (res.config.var[1], ...))` continues from the trained map (docs/src/index.md:129)
            prev = getattr(lf, "_engine", None)
            trained = prev is not None and prev is not self and getattr(prev, "p", None)
            if isinstance(lf, ContinuousVar):
                d.kind, d.npoints = _lib.CONTINUOUS, lf.ninc
                init = lf.grid if trained else lf._grid0
            elif isinstance(lf, FermiK):
                d.kind, d.npoints = _lib.FERMIK, lf.dim   # lower = kF, upper = dk, alpha = maxK
            else:
                d.kind, d.npoints = _lib.DISCRETE, 0
                init = lf.distribution if trained else lf._dist0
            if init is not None:
                init = np.ascontiguousarray(init, dtype=np.float64)
                self._keep.append(init)
                d.init = _dp(init)
        dof = np.ascontiguousarray(config.dof, dtype=np.int32)
        onb = np.ascontiguousarray(config.obs_nbin, dtype=np.int32)
        obd = np.ascontiguousarray(config.obs_bin_draw(measure), dtype=np.int32)
        nb_off = nb_list = None
        nbrs = config.neighbor_lists()   # None = the reference default (configuration.jl:203-208)
        if nbrs is not None:
            off = np.zeros(len(nbrs) + 1, dtype=np.int32)
            off[1:] = np.cumsum([len(n) for n in nbrs])
            flat = np.ascontiguousarray([j for n in nbrs for j in n], dtype=np.int32)
            self._keep += [off, flat]
            nb_off, nb_list = off.ctypes.data_as(c_int32_p), flat.ctypes.data_as(c_int32_p)
        desc = ProblemDesc(len(leaves), arr, len(config.var), config.N, dof.ctypes.data_as(c_int32_p),
                           onb.ctypes.data_as(c_int32_p), obd.ctypes.data_as(c_int32_p), nb_off, nb_list, config.ncomp)
        self.p = C.c_void_p()
        check(L.mci_problem_create(self.ctx, C.byref(desc), C.byref(self.p)))
        ud = integrand.userdata
        if isinstance(integrand, HostIntegrand) and getattr(integrand, "indexed", False):
            self._host_cb = _lib.HOST_INTEGRAND_IDX_FN(self._make_host_indexed_callback(integrand.fn))   # keep alive
            check(L.mci_set_integrand_host_indexed(self.p, C.cast(self._host_cb, C.c_void_p), None))
        elif isinstance(integrand, HostIntegrand):
            self._host_cb = _lib.HOST_INTEGRAND_FN(self._make_host_callback(integrand.fn, getattr(integrand, "inplace", False)))   # keep alive
            check(L.mci_set_integrand_host(self.p, C.cast(self._host_cb, C.c_void_p), None))
        else:
            check(L.mci_set_integrand_source(self.p, integrand.body.encode(), _dp(ud) if len(ud) else None, len(ud)))
        if isinstance(measure, Measure):
            check(L.mci_set_measure_source(self.p, measure.body.encode()))
        elif isinstance(measure, HostMeasure) and measure.indexed:
            self._host_measure_cb = _lib.HOST_MEASURE_IDX_FN(self._make_host_measure_indexed_callback(measure.fn))   # keep alive
            check(L.mci_set_measure_host_indexed(self.p, C.cast(self._host_measure_cb, C.c_void_p), None))
        elif isinstance(measure, HostMeasure):
            self._host_measure_cb = _lib.HOST_MEASURE_FN(self._make_host_measure_callback(measure.fn))   # keep alive
            check(L.mci_set_measure_host(self.p, C.cast(self._host_measure_cb, C.c_void_p), None))
        if threads or wg_per_block is not None:
            check(L.mci_set_launch(self.p, threads or 0, -1 if wg_per_block is None else wg_per_block))
        if rng_bits is not None:
            check(L.mci_set_rng_bits(self.p, int(rng_bits)))
        if rng_rounds is not None:
            check(L.mci_set_rng_rounds(self.p, int(rng_rounds)))
        if deterministic:
            check(L.mci_set_deterministic(self.p, 1))
        nd, no, ps, tm, lds = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64()
        check(L.mci_problem_info(self.p, C.byref(nd), C.byref(no), C.byref(ps), C.byref(tm), C.byref(lds)))
        self.ndraw, self.nobs, self.packed_size, self.table_mode, self.lds_bytes = nd.value, no.value, ps.value, tm.value, lds.value
        if device >= 0 and not np.allclose(config._reweight0, 1.0 / (config.N + 1)):
            r = np.ascontiguousarray(config._reweight0)
            check(L.mci_set_reweight(self.p, _dp(r), len(r)))
        for i, lf in enumerate(leaves):
            lf._engine, lf._leaf_index = self, i

    def _make_host_callback(self, fn, inplace=False):
        """ctypes trampoline: draw-major x[k*n + i] -> numpy views -> fn(x, config) -> w[q*n + i]; inplace: fn(x, weights, config)
        writes into a view of w itself (vegas/montecarlo.jl:140-141).  A closure written per sample with Python branches on its draws
        (`1.0 if x[0] ** 2 + x[1] ** 2 < 1 else 0.0`: the reference's own style) cannot take a batch -- numpy refuses the truth value
        of an array -- and is then called sample by sample (slow; such a closure normally never gets here: the tracer writes its
        branches out as selects)."""
        config = self.config
        nc = config.ncomp
        state = {"per_sample": False}

        def views(X, n):
            return self._pool_views(X, n)

        def batch(X, W, n):
            arg = views(X, n)
            if inplace:
                if nc == 2:
                    Z = np.zeros((config.N, n), dtype=complex)
                    fn(arg, Z, config)
                    W[0::2], W[1::2] = Z.real, Z.imag
                else:
                    W[:] = 0.0
                    fn(arg, W, config)
                return
            out = fn(arg, config)
            if config.N == 1 and not isinstance(out, (tuple, list)):
                out = (out,)
            assert len(out) == config.N, "the integrand must return one value per integrand"
            for i, o in enumerate(out):
                o = np.broadcast_to(np.asarray(o), (n,))
                if nc == 2:
                    W[2 * i] = o.real
                    W[2 * i + 1] = o.imag
                else:
                    W[i] = o

        def per_sample(X, W, n):
            Z = np.zeros((config.N, n), dtype=complex if nc == 2 else float)
            for j in range(n):
                arg = self._pool_views(X[:, j:j + 1], 1, scalar=True)
                if inplace:
                    w = np.zeros(config.N, dtype=Z.dtype)
                    fn(arg, w, config)
                    Z[:, j] = w
                else:
                    out = fn(arg, config)
                    if config.N == 1 and not isinstance(out, (tuple, list)):
                        out = (out,)
                    assert len(out) == config.N, "the integrand must return one value per integrand"
                    for i, o in enumerate(out):
                        Z[i, j] = o
            if nc == 2:
                W[0::2], W[1::2] = Z.real, Z.imag
            else:
                W[:] = Z

        def cb(xp, wp, n, ndraw, nw, user):
            try:
                X = np.ctypeslib.as_array(xp, shape=(ndraw, n))
                W = np.ctypeslib.as_array(wp, shape=(nw, n))
                if not state["per_sample"]:
                    try:
                        batch(X, W, n)
                        return 0
                    except (ValueError, TypeError, IndexError) as e:
                        # a closure written per sample like the reference's: a Python branch on a draw ("The truth value of an array
                        # with more than one element is ambiguous"), a list indexed with a Discrete draw ("only integer scalar arrays
                        # can be converted to a scalar index"), ...: sample by sample from here on (an error of its own comes again)
                        if n <= 1:
                            raise
                        _slow_path_warning("integrand", "sample", e)
                        state["per_sample"] = True
                per_sample(X, W, n)
                return 0
            except Exception:   # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        return cb

    def _make_host_indexed_callback(self, fn):
        """ctypes trampoline of the `integrand(idx, var, config)` form: one call of fn per integrand index some chain asks for,
        over the chains that ask for it"""
        config = self.config
        nc = config.ncomp
        state = {"per_sample": False}

        def cb(ip, xp, wp, n, ndraw, ncomp, user):
            try:
                idx = np.ctypeslib.as_array(ip, shape=(n,))
                X = np.ctypeslib.as_array(xp, shape=(ndraw, n))
                W = np.ctypeslib.as_array(wp, shape=(ncomp, n))
                for i in np.unique(idx[idx >= 0]):
                    sel = np.nonzero(idx == i)[0]
                    whole = len(sel) == n
                    Xs = X if whole else np.ascontiguousarray(X[:, sel])
                    o = None
                    if not state["per_sample"]:
                        try:
                            o = fn(int(i), self._pool_views(Xs, len(sel)), config)
                        except (ValueError, TypeError, IndexError) as e:   # a closure written per sample: sample by sample (see _make_host_callback)
                            if len(sel) <= 1:
                                raise
                            _slow_path_warning("integrand", "sample", e)
                            state["per_sample"] = True
                    if o is None:
                        o = np.array([fn(int(i), self._pool_views(np.ascontiguousarray(Xs[:, j:j + 1]), 1, scalar=True), config) for j in range(len(sel))])
                    o = np.broadcast_to(np.asarray(o), (len(sel),))
                    if nc == 2:
                        W[0, sel], W[1, sel] = o.real, o.imag
                    else:
                        W[0, sel] = o
                return 0
            except Exception:   # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        return cb

    def _pool_views(self, X, n, scalar=False):
        """draw-major [ndraw, n] -> what the reference hands a closure: the pool itself with one variable type, else a tuple of pools
        (scalar: one sample without the batch axis, for a closure that is called sample by sample).  A Discrete pool holds integers like
        the reference's (variable.jl:283: `grid[bin[0] - 1]`, `obs[0][ext[0] - 1] += w` index with them as they are); a CompositeVar
        is indexed [leaf][slot] (variable.jl:436-447: `x, y = cvar`) -- one array when its leaves are of one kind, else a tuple of
        per-leaf arrays; a pool with `offset` gets that many leading zero slots: the reference's closures address X[i + offset]
        (variable.jl:577, test/montecarlo.jl:19-32)."""
        layout = self.config.pool_layout()

        def typed(a, kind):
            return np.rint(a).astype(np.int64) if kind == "d" else a

        def pool(k0, md, nl, off, composite, kinds):
            a = X[k0:k0 + md * nl].reshape((md, n) if nl == 1 and not composite else (md, nl, n))
            if off:
                a = np.concatenate([np.zeros((off,) + a.shape[1:]), a])
            if composite:                  # [leaf][slot]
                a = a.transpose(1, 0, 2)
                if len(set(kinds)) > 1:
                    return tuple(typed(a[l][..., 0] if scalar else a[l], kinds[l]) for l in range(nl))
            return typed(a[..., 0] if scalar else a, kinds[0])
        if len(layout) == 1 and layout[0][2:] == (1, 0, False, "c"):
            return X[:, 0] if scalar else X
        arg = tuple(pool(*p) for p in layout)
        return arg[0] if len(arg) == 1 else arg

    def _make_host_measure_callback(self, fn):
        """ctypes trampoline: one block's draws and relative weights -> numpy views -> fn(x, obs, weights, config) -> obs[nobs].
        The closure is called once with the block's records as arrays (the batch form: `obs[0][0] += weights[0].sum()`); one written
        per record like the reference's (`obs[1][1] += weights[1]`, vegas/montecarlo.jl:156-161) raises on arrays and is then called
        record by record.  `obs[i][bins] += w` with an integer ARRAY of bins (a Discrete draw over the records) accumulates every
        record, repeated bins included (_HostObs)."""
        config = self.config
        nc = config.ncomp
        dt = complex if nc == 2 else float
        state = {"per_record": False}

        def cb(xp, rp, n, stride, ndraw, nw, block, op, nobs, user):
            try:
                X = np.ctypeslib.as_array(xp, shape=(ndraw, stride))[:, :n]
                R = np.ctypeslib.as_array(rp, shape=(nw, stride))[:, :n]
                O = np.ctypeslib.as_array(op, shape=(nobs,))
                weights = [R[2 * i] + 1j * R[2 * i + 1] for i in range(config.N)] if nc == 2 else [R[i] for i in range(config.N)]
                obs = None
                if not state["per_record"]:
                    obs = [_HostObs(shape, dt) for shape in config.obs_shape]
                    try:
                        fn(self._pool_views(X, n), obs, weights, config)
                    except (ValueError, TypeError, IndexError) as e:
                        if n <= 1:
                            raise
                        _slow_path_warning("measure", "record", e)
                        state["per_record"], obs = True, None
                if obs is None:
                    obs = [np.zeros(shape, dtype=dt) for shape in config.obs_shape]
                    for j in range(n):
                        fn(self._pool_views(X[:, j:j + 1], 1, scalar=True), obs, [w[j] for w in weights], config)
                _store_obs(O, obs, config.obs_nbin, nc)
                return 0
            except Exception:   # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        return cb

    def _make_host_measure_indexed_callback(self, fn):
        """ctypes trampoline of the `measure(idx, var, obs, relative_weight, config)` form: one call of fn per integrand index that
        some record of the block belongs to, over those records (record by record for a closure that raises on arrays, see above)"""
        config = self.config
        nc = config.ncomp
        dt = complex if nc == 2 else float
        state = {"per_record": False}

        def cb(ip, xp, rp, n, stride, ndraw, ncomp, block, op, nobs, user):
            try:
                idx = np.ctypeslib.as_array(ip, shape=(n,))
                X = np.ctypeslib.as_array(xp, shape=(ndraw, stride))[:, :n]
                R = np.ctypeslib.as_array(rp, shape=(ncomp, stride))[:, :n]
                O = np.ctypeslib.as_array(op, shape=(nobs,))
                obs, off = [], 0
                for shape, nb in zip(config.obs_shape, config.obs_nbin):   # the block's observables so far (the library calls once per integrand)
                    o = np.array(O[off:off + nb])
                    obs.append(((o[0::2] + 1j * o[1::2]) if nc == 2 else o.astype(dt)).reshape(shape).view(_HostObs))
                    off += nb
                w = (R[0] + 1j * R[1]) if nc == 2 else R[0]
                for i in np.unique(idx[idx >= 0]):
                    sel = np.nonzero(idx == i)[0]
                    whole = len(sel) == n
                    Xs = X if whole else np.ascontiguousarray(X[:, sel])
                    ws = w if whole else w[sel]
                    if not state["per_record"]:
                        before = [np.array(o) for o in obs]
                        try:
                            fn(int(i), self._pool_views(Xs, len(sel)), obs, ws, config)
                            continue
                        except (ValueError, TypeError, IndexError) as e:
                            if len(sel) <= 1:
                                raise
                            _slow_path_warning("measure", "record", e)
                            state["per_record"] = True
                            for o, q in zip(obs, before):
                                np.ndarray.__setitem__(o, slice(None), q)
                    plain = [o.view(np.ndarray) for o in obs]
                    for j in range(len(sel)):
                        fn(int(i), self._pool_views(np.ascontiguousarray(Xs[:, j:j + 1]), 1, scalar=True), plain, ws[j], config)
                _store_obs(O, obs, config.obs_nbin, nc)
                return 0
            except Exception:   # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        return cb

    def close(self):
        if getattr(self, "p", None):
            lib().mci_problem_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- kernels -------------------------------------------------------------------------------
    @staticmethod
    def _kernel_code(solver):
        return {"vegas_persistent": 3, "vegasmc_lanes": 5, "mcmc_lanes": 6}.get(solver) or _lib.SOLVERS[solver]

    def compile(self, solver="vegas"):
        """solver: "vegas" | "vegasmc" | "mcmc" | "vegas_persistent" (the persistent :vegas kernel, layouts with one Continuous leaf) |
        "vegasmc_lanes" | "mcmc_lanes" (the chain solvers' kernels with several lanes per chain, csrc/mci_spec.h)"""
        check(lib().mci_compile_solver(self.p, self._kernel_code(solver)))

    def code_object(self, solver="vegas"):
        """kernel-cache file holding the solver's gfx950 code object (after compile / the first run)"""
        buf = C.create_string_buffer(4096)
        check(lib().mci_kernel_code_object(self.p, self._kernel_code(solver), buf, len(buf)))
        return buf.value.decode()

    def set_kernel_timing(self, mode=-1):
        """HIP events around every sample launch (kernel_times_ms): -1 launches of >= 2^20 samples (default), 0 never, 1 always"""
        check(lib().mci_set_kernel_timing(self.p, int(mode)))

    def histogram_copies(self):
        """interleaved copies of the LDS histograms in the :vegas sample kernel (1 = the plain layout)"""
        n = C.c_int32()
        check(lib().mci_get_histogram_copies(self.p, C.byref(n)))
        return n.value

    def check_status(self):
        """synchronise and raise what the device flagged (normalization / histogram / chain start errors)"""
        check(lib().mci_check_status(self.p))

    def set_launch(self, threads=0, wg_per_block=-1):
        check(lib().mci_set_launch(self.p, threads, wg_per_block))

    def run(self, solver, nevalperblock, block_lo, block_hi, iteration, seed, measurefreq=1, nchain=0, thermal_ratio=0.1):
        check(lib().mci_iteration_run(self.p, _lib.SOLVERS[solver], int(nevalperblock), int(block_lo), int(block_hi),
                                      int(iteration), int(seed), int(measurefreq), int(nchain), float(thermal_ratio)))

    def reduce(self):
        check(lib().mci_iteration_reduce(self.p))

    def finish(self, solver, block_total, adapt=True, gamma=1.0, want_stats=True):
        if not want_stats:
            check(lib().mci_iteration_finish(self.p, _lib.SOLVERS[solver], int(block_total), 1 if adapt else 0, gamma, None, None))
            return None, None
        m, e = np.empty(self.nobs), np.empty(self.nobs)
        check(lib().mci_iteration_finish(self.p, _lib.SOLVERS[solver], int(block_total), 1 if adapt else 0, gamma, _dp(m), _dp(e)))
        return m, e

    def iteration(self, solver, nevalperblock, block_lo, block_hi, iteration, seed, measurefreq=1, nchain=0, thermal_ratio=0.1):
        """run + read back the local packed buffer [obsSum|obsSqSum|normalization|neval|visited|histograms]"""
        self.run(solver, nevalperblock, block_lo, block_hi, iteration, seed, measurefreq, nchain, thermal_ratio)
        return self.get_packed()

    def iteration_log(self, nrows):
        """statistics head of the last nrows finished iterations, [nrows, 2*nobs+2+N+1] (oldest first)"""
        nstat = 2 * self.nobs + 2 + self.config.N + 1
        out = np.empty((nrows, nstat))
        check(lib().mci_get_iteration_log(self.p, int(nrows), _dp(out)))
        return out

    def reserve_iterations(self, rows):
        """room for `rows` more iterations in the device-side log, so that none of them synchronises to grow it"""
        check(lib().mci_reserve_iteration_log(self.p, int(rows)))

    def get_packed(self, reduce_size=False):
        """the packed buffer [statistics | histograms | propose | accept]; reduce_size: with the 64 :mcmc holding-time counts behind it --
        what an external reducer sums over the ranks (mci_reduce_size)"""
        out = np.empty(self.reduce_size if reduce_size else self.packed_size)
        check(lib().mci_get_packed(self.p, _dp(out), len(out)))
        return out

    @property
    def reduce_size(self):
        return self.packed_size + 64

    def external_reduce_done(self):
        """an external reducer has summed reduce_size doubles of the packed buffer over the ranks; see mci_external_reduce_done"""
        check(lib().mci_external_reduce_done(self.p))

    def comm_collectives(self):
        """(ncclAllReduce calls the library has issued on this engine's context, element count of the last one)"""
        n, c = C.c_int64(), C.c_int64()
        check(lib().mci_comm_collectives(context(self.device), C.byref(n), C.byref(c)))
        return int(n.value), int(c.value)

    def set_packed(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        check(lib().mci_set_packed(self.p, _dp(a), len(a)))

    def packed_device_ptr(self):
        return lib().mci_packed_device_ptr(self.p)

    def hold_histogram(self):
        """:mcmc: chains of the last launch by bit_width(longest holding time) -- what the automatic chain length follows"""
        out = (C.c_uint64 * 64)()
        check(lib().mci_get_hold_histogram(self.p, out))
        return np.array(list(out), dtype=np.uint64)

    def stream(self):
        """the library's hipStream_t (as an integer), for callers that order their own work with ours"""
        return lib().mci_ctx_stream(context(self.device)) or 0

    def save_state(self, path):
        """grids, distributions and reweight -> MCISTATE file (resume across processes)"""
        check(lib().mci_save_state(self.p, str(path).encode()))

    def load_state(self, path):
        check(lib().mci_load_state(self.p, str(path).encode()))

    def train(self):
        check(lib().mci_train(self.p))

    def set_rng_rounds(self, rounds):
        """Philox4x32 rounds of every stream: 10 (default) or 7 (opt-in, 30 % less generator work); see mci_set_rng_rounds"""
        check(lib().mci_set_rng_rounds(self.p, int(rounds)))

    def set_rng_bits(self, bits):
        """:vegas sample stream: 52 (default, the resolution of rand(Float64)) or 32 random bits per draw; see mci_set_rng_bits"""
        check(lib().mci_set_rng_bits(self.p, int(bits)))

    def set_deterministic(self, on=True):
        """bit-identical results for a fixed seed, run to run (one LDS histogram copy per wave, fixed merge orders); see
        mci_set_deterministic"""
        check(lib().mci_set_deterministic(self.p, 1 if on else 0))

    def set_chain_carry(self, mode):
        """chain solvers with many chains per block: "auto" (default) / "on" -- the next iteration of the same solver over the same
        blocks continues the chains of the previous one (:mcmc: resampled to the reweight factors doReweight! has just moved) -- or
        "off" (every launch draws new starts and burns them in); see mci_set_chain_carry"""
        check(lib().mci_set_chain_carry(self.p, {"auto": -1, "off": 0, "on": 1, -1: -1, 0: 0, 1: 1}[mode]))

    def set_iteration_counted(self, counted):
        """for a loop over run / reduce / finish (integrate() with several ranks): the launches that follow enter the final average
        (iteration >= ignore); see mci_set_iteration_counted"""
        check(lib().mci_set_iteration_counted(self.p, 1 if counted else 0))

    def set_persistent(self, mode):
        """launch-bound :vegas calls of integrate() as ONE persistent launch: "auto" (default), "off" or "on" (whenever the layout
        allows, whatever the size); see mci_set_persistent"""
        check(lib().mci_set_persistent(self.p, {"auto": -1, "off": 0, "on": 1, -1: -1, 0: 0, 1: 1}[mode]))

    def last_integrate_persistent(self):
        """did the last integrate() of this engine run as one persistent launch?"""
        v = C.c_int32()
        check(lib().mci_last_integrate_persistent(self.p, C.byref(v)))
        return bool(v.value)

    def last_chain_launch(self):
        """(chains per block, continued the previous launch?) of the last chain-solver launch"""
        n, c = C.c_int64(), C.c_int32()
        check(lib().mci_last_chain_launch(self.p, C.byref(n), C.byref(c)))
        return int(n.value), bool(c.value)

    def set_chain_speculation(self, lanes=-1, accept=0.0, max_accepts=-1):
        """several lanes per chain (csrc/mci_spec.h): lanes -1 automatic, 1 never, 2..64 forced; the acceptance the speculation tree
        is built for and the most accept edges on a way through it (<= 0 / < 0: the solver's defaults); see mci_set_chain_speculation"""
        check(lib().mci_set_chain_speculation(self.p, int(lanes), float(accept), int(max_accepts)))

    def last_chain_speculation(self):
        """(lanes per chain, accept levels of the tree) of the last chain-solver launch; (1, 0) = one lane per chain"""
        g, m = C.c_int32(), C.c_int32()
        check(lib().mci_last_chain_speculation(self.p, C.byref(g), C.byref(m)))
        return int(g.value), int(m.value)

    def chain_speculation_status(self, solver):
        """the self-check of the solver's several-lanes-per-chain code object: 0 not launched yet, 1 verified against the lane-per-chain
        kernel (now or by an earlier process: the marker in the kernel cache), -1 failed (this problem keeps one lane per chain), -2 did
        not compile; see mci_chain_speculation_status"""
        st = C.c_int32()
        check(lib().mci_chain_speculation_status(self.p, _lib.SOLVERS[solver], C.byref(st)))
        return int(st.value)

    def compile_chain_speculation(self, solver):
        """the several-lanes-per-chain kernel of "vegasmc" | "mcmc" (its own code object)"""
        check(lib().mci_compile_chain_speculation(self.p, _lib.SOLVERS[solver]))

    def set_train_walk(self, mode):
        """train!'s refinement walk: "serial" (the reference's recurrence), "scan" (prefix scan + bisection), "auto", or "serial_general"
        (the recurrence without the predicted decisions: what "serial" falls back to); see mci_set_train_walk.  "serial_wrong_decision" is
        the test hook of csrc/mci_debug.h: "serial" with one decision planted wrong, so that its check and the fall-back run"""
        check(lib().mci_debug_plant_wrong_decision(self.p, 1 if mode in ("serial_wrong_decision", 3) else 0))
        check(lib().mci_set_train_walk(self.p, {"auto": -1, "scan": 0, "serial": 1, "serial_general": 2, "serial_wrong_decision": 1, -1: -1, 0: 0, 1: 1, 2: 2, 3: 1}[mode]))

    def walk_counts(self):
        """(serial walks of train! run as slots with given decisions, walks in the general form) so far; see mci_debug_walk_counts"""
        out = (C.c_int64 * 2)()
        check(lib().mci_debug_walk_counts(self.p, out))
        return int(out[0]), int(out[1])

    def split_chunks(self):
        """(chunks, bytes of parked stream held at a time) of the last many-grid :vegas launch; see csrc/mci_debug.h mci_debug_split_chunks"""
        n, b = C.c_int64(), C.c_int64()
        check(lib().mci_debug_split_chunks(self.p, C.byref(n), C.byref(b)))
        return int(n.value), int(b.value)

    def integrate(self, solver, neval, niter=10, block=16, ignore=-1, adapt=True, gamma=1.0, measurefreq=1, seed=1234,
                  nchain=0, first_iteration=0, thermal_ratio=0.1, reweight_goal=None):
        """the whole loop inside the library (mci_integrate)"""
        goal = None
        if reweight_goal is not None:
            self._goal = np.ascontiguousarray(reweight_goal, dtype=np.float64)
            goal = _dp(self._goal)
        a = _lib.IntegrateArgs(_lib.SOLVERS[solver], int(neval), int(niter), int(block), int(ignore), 1 if adapt else 0,
                               float(gamma), int(measurefreq), int(seed), int(nchain), int(first_iteration),
                               float(thermal_ratio), goal)
        n = self.nobs
        im, ie = np.zeros((niter, n)), np.zeros((niter, n))
        m, s, c2 = np.zeros(n), np.zeros(n), np.zeros(n)
        vis = np.zeros(self.config.N + 1)
        r = _lib.ResultC(niter, n, _dp(im), _dp(ie), _dp(m), _dp(s), _dp(c2), 0, 0.0, _dp(vis), 0, 0)
        check(lib().mci_integrate(self.p, C.byref(a), C.byref(r)))
        dn, dl = C.c_int64(), C.c_int32()
        check(lib().mci_last_integrate_discarded(self.p, C.byref(dn), C.byref(dl)))
        out = dict(mean=m, stdev=s, chi2=c2, iter_mean=im, iter_std=ie, neval=r.neval, seconds=r.seconds, visited=vis,
                   correlated=bool(r.correlated), block_mean=None, warmup=int(r.warmup), neval_discarded=int(dn.value))
        if _lib.SOLVERS[solver] != _lib.VEGAS and out["correlated"]:
            # (only a run of carried chains needs them -- the block-lineage error of Result.with_ignore; a launch-bound default-size call is
            # spared the flush, the copy and the second synchronisation)
            out["block_mean"] = self.block_means(niter)[0]
        return out

    def mcmc_launch_valid(self):
        """(the last automatic :mcmc launch was long enough for the holds it measured, some launch of this problem has been, its chain
        length, its longest hold); waits for that launch's sample kernel -- see mci_mcmc_launch_valid"""
        v, w, ln, h = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        check(lib().mci_mcmc_launch_valid(self.p, C.byref(v), C.byref(w), C.byref(ln), C.byref(h)))
        return bool(v.value), bool(w.value), int(ln.value), int(h.value)

    def discard_iteration(self):
        """the last finished iteration out of the iteration log and the block log again (a warm-up launch that is run again)"""
        check(lib().mci_iteration_discard(self.p))

    def block_means(self, rows):
        """chain solvers: (block means [rows][local blocks][nobs] of the last `rows` iterations, how many of the logged iterations
        continued the chains of the one before); see mci_get_block_means"""
        nb, carried = C.c_int64(), C.c_int32()
        check(lib().mci_get_block_means(self.p, 0, None, C.byref(nb), C.byref(carried)))
        out = np.zeros((int(rows), int(nb.value), self.nobs))
        if rows > 0 and out.size:
            check(lib().mci_get_block_means(self.p, int(rows), _dp(out), C.byref(nb), C.byref(carried)))
        return out, int(carried.value)

    def comm_sum(self, v):
        """v summed over the ranks of the library's communicator (mci_comm_sum)"""
        a = np.ascontiguousarray(v, dtype=np.float64).copy()
        check(lib().mci_comm_sum(self.p, _dp(a), a.size))
        return a

    def reset_block_log(self):
        check(lib().mci_reset_block_log(self.p))

    def comm_ranks(self):
        """ranks of the communicator attached to this engine's context (1: none; mci_integrate shards its blocks over them)"""
        r, n = C.c_int32(), C.c_int32()
        check(lib().mci_comm_rank(context(self.device), C.byref(r), C.byref(n)))
        return int(n.value)

    def sample_dump(self, n, nevalperblock=None, block_index=0, iteration=0, seed=1234):
        nevalperblock = n if nevalperblock is None else nevalperblock
        x, jac, w = np.empty((n, self.ndraw)), np.empty(n), np.empty((n, self.config.N * self.config.ncomp))
        check(lib().mci_sample_dump(self.p, iteration, seed, int(nevalperblock), int(block_index), int(n), _dp(x), _dp(jac), _dp(w)))
        return x, jac, w

    def kernel_times_ms(self, n=512):
        """HIP-event durations of the last n sampling-kernel launches (oldest first), (workgroups, threads)"""
        ms = (C.c_float * n)()
        got, wg, th = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().mci_kernel_times_ms(self.p, ms, n, C.byref(got), C.byref(wg), C.byref(th)))
        return np.array(ms[:got.value], dtype=np.float64), wg.value, th.value

    def kernel_clocks_mhz(self, n=512):
        """shader clock (MHz) of the last n timed :vegas launches, from s_memtime / s_memrealtime around the sample loop; see mci_kernel_clocks"""
        out = np.empty(n)
        got = C.c_int32()
        check(lib().mci_kernel_clocks(self.p, _dp(out), n, C.byref(got)))
        return out[:got.value].copy()

    def comm_times_ms(self, n=64):
        """HIP-event durations of this rank's last n per-iteration all-reduces inside the library (oldest first; recorded under the
        sample launch's timing rule, set_kernel_timing)"""
        ms = (C.c_float * n)()
        got = C.c_int32()
        check(lib().mci_comm_times_ms(self.p, ms, n, C.byref(got)))
        return np.array(ms[:got.value], dtype=np.float64)

    def last_kernel_ms(self):
        ms, wg, th = self.kernel_times_ms(1)
        return (float(ms[-1]) if len(ms) else float("nan")), wg, th   # (nan: the launch ran without events, set_kernel_timing)

    # ---- state ---------------------------------------------------------------------------------
    def grid(self, leaf):
        n = self.config.leaves[leaf].ninc
        out = np.empty(n)
        check(lib().mci_get_grid(self.p, leaf, _dp(out), n))
        return out

    def set_grid(self, leaf, grid):
        g = np.ascontiguousarray(grid, dtype=np.float64)
        check(lib().mci_set_grid(self.p, leaf, _dp(g), len(g)))

    def distribution(self, leaf):
        lf = self.config.leaves[leaf]
        k = lf.upper - lf.lower + 1
        d, a = np.empty(k), np.empty(k + 1)
        check(lib().mci_get_distribution(self.p, leaf, _dp(d), _dp(a), k))
        return d, a

    def set_distribution(self, leaf, dist):
        d = np.ascontiguousarray(dist, dtype=np.float64)
        check(lib().mci_set_distribution(self.p, leaf, _dp(d), len(d)))

    def histogram(self, leaf):
        """histogram section of the packed buffer for one leaf"""
        packed = self.get_packed()
        off = 2 * self.nobs + 2 + self.config.N + 1
        for i, lf in enumerate(self.config.leaves):
            nb = (lf.ninc - 1) if isinstance(lf, ContinuousVar) else 1 if isinstance(lf, FermiK) else (lf.upper - lf.lower + 1)
            if i == leaf:
                return packed[off:off + nb]
            off += nb
        raise IndexError(leaf)

    def reweight(self):
        out = np.empty(self.config.N + 1)
        check(lib().mci_get_reweight(self.p, _dp(out), len(out)))
        return out

    def acceptance(self):
        """(propose, accept) of the last iteration, each shaped [3, N+1, max(N+1, Nv)] like config.propose / config.accept
        (configuration.jl:185-186; 0-based: [update, integrand, target]); see mci_get_acceptance"""
        nd = self.config.N + 1
        m = max(nd, len(self.config.var))
        n = 3 * nd * m
        pr, ac = np.empty(n), np.empty(n)
        check(lib().mci_get_acceptance(self.p, _dp(pr), _dp(ac), n))
        return pr.reshape(3, nd, m), ac.reshape(3, nd, m)

    def histograms(self):
        """histogram section of the packed buffer (all leaves, concatenated)"""
        off = 2 * self.nobs + 2 + self.config.N + 1
        nd = self.config.N + 1
        npa = 3 * nd * max(nd, len(self.config.var))
        return self.get_packed()[off:self.packed_size - 2 * npa]

    def set_reweight_goal(self, goal):
        if goal is None:
            check(lib().mci_set_reweight_goal(self.p, None, 0))
        else:
            g = np.ascontiguousarray(goal, dtype=np.float64)
            check(lib().mci_set_reweight_goal(self.p, _dp(g), len(g)))

    def set_reweight(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        check(lib().mci_set_reweight(self.p, _dp(r), len(r)))
